#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_v2xvit.py tests/test_shell.py -m gpu -q -x -s > gpurun_out/r2c24_pytest.log 2>&1
rc=$?
grep -h "worst relative error\|directional" gpurun_out/r2c24_pytest.log | cut -c1-200
tail -n 30 gpurun_out/r2c24_pytest.log | cut -c1-300
echo "pytest rc=$rc"
