#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python tools/diag_r2.py > gpurun_out/r2c2_diag.log 2>&1
rc=$?
tail -n 60 gpurun_out/r2c2_diag.log
if [ $rc -ne 0 ]; then echo "diag rc=$rc"; exit $rc; fi
timeout -k 10 600 python -m pytest tests/test_gpu_conv8.py tests/test_gpu_parity.py -m gpu -q -s -x > gpurun_out/r2c2_pytest.log 2>&1
rc=$?
tail -n 30 gpurun_out/r2c2_pytest.log
echo "pytest rc=$rc"
