#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_pillars.py tests/test_gpu_backward.py -m gpu -q -x -k "pillar or stage1 or stage2" > gpurun_out/r2c32_pytest.log 2>&1
rc=$?
tail -n 25 gpurun_out/r2c32_pytest.log | cut -c1-300
echo "pytest rc=$rc"
