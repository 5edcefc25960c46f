#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "half_precision" > gpurun_out/r2c28_pytest.log 2>&1
rc=$?
tail -n 15 gpurun_out/r2c28_pytest.log | cut -c1-300
echo "pytest rc=$rc"
