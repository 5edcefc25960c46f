#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_v2xvit.py -m gpu -q -x > gpurun_out/r2c10_pytest.log 2>&1
rc=$?
tail -n 40 gpurun_out/r2c10_pytest.log | cut -c1-260
echo "pytest rc=$rc"
