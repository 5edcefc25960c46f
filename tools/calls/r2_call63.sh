#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_backward.py -m gpu -q -x > gpurun_out/r2c63_pytest.log 2>&1
rc=$?
tail -n 4 gpurun_out/r2c63_pytest.log | cut -c1-250
echo "pytest rc=$rc"
