#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py tests/test_shell.py -m gpu -q -x -s -k "message_extractor or stage2" > gpurun_out/r2c22_pytest.log 2>&1
rc=$?
grep -h "worst relative error\|stage-2 training" gpurun_out/r2c22_pytest.log | cut -c1-200
tail -n 25 gpurun_out/r2c22_pytest.log | cut -c1-300
echo "pytest rc=$rc"
