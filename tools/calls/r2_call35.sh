#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_shell.py tests/test_gpu_backward.py -m gpu -q -x -k "where2comm" > gpurun_out/r2c35_pytest.log 2>&1
rc=$?
tail -n 40 gpurun_out/r2c35_pytest.log | cut -c1-300
echo "pytest rc=$rc"
