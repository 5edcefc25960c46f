#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_backbone.py -m gpu -q -x > gpurun_out/r2c47_pytest.log 2>&1
rc=$?
tail -n 25 gpurun_out/r2c47_pytest.log | cut -c1-300
echo "pytest rc=$rc"
