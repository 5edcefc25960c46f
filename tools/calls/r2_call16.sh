#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_parity.py -m gpu -q -x -s > gpurun_out/r2c16_pytest.log 2>&1
rc=$?
grep -h "worst relative error" gpurun_out/r2c16_pytest.log | cut -c1-200
tail -n 25 gpurun_out/r2c16_pytest.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/train_bench.py > gpurun_out/r2c16_train.log 2>&1 || { tail gpurun_out/r2c16_train.log; exit 1; }
cat gpurun_out/r2c16_train.log
