#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py -m gpu -q -x -s -k "training_loop" > gpurun_out/r2c31_pytest.log 2>&1
rc=$?
grep -h "generation loss" gpurun_out/r2c31_pytest.log | cut -c1-200
tail -n 12 gpurun_out/r2c31_pytest.log | cut -c1-300
echo "pytest rc=$rc"
