#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_conv8.py --deselect tests/test_gpu_properties.py --deselect tests/test_gpu_configs.py > gpurun_out/r2c66_pytest.log 2>&1
rc=$?
tail -n 4 gpurun_out/r2c66_pytest.log | cut -c1-250
echo "pytest rc=$rc"
