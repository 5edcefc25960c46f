#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r2c43_pytest.log 2>&1
rc=$?
tail -n 5 gpurun_out/r2c43_pytest.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/v2xvit_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c43_v2xvit.log
timeout -k 10 300 python tools/train_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c43_train.log
