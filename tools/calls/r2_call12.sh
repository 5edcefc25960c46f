#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_second.py -m gpu -q -x -s > gpurun_out/r2c12_pytest.log 2>&1
rc=$?
tail -n 40 gpurun_out/r2c12_pytest.log | cut -c1-300
echo "pytest rc=$rc"
