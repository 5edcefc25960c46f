#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_second.py -m gpu -q -x -s > gpurun_out/r2c30_pytest.log 2>&1
rc=$?
grep -h "worst relative\|SECOND HIP" gpurun_out/r2c30_pytest.log | cut -c1-200
tail -n 25 gpurun_out/r2c30_pytest.log | cut -c1-300
echo "pytest rc=$rc"
