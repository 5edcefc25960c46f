#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py tests/test_pillars.py tests/test_shell.py -m gpu -q -x -s -k "stage1 or stage2 or pillar or shell" > gpurun_out/r2c27_pytest.log 2>&1
rc=$?
grep -h "stage-1 training\|stage-2 training" gpurun_out/r2c27_pytest.log | cut -c1-400
tail -n 25 gpurun_out/r2c27_pytest.log | cut -c1-300
echo "pytest rc=$rc"
