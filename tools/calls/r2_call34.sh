#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_where2comm.py tests/test_v2xvit.py -m gpu -q -x > gpurun_out/r2c34_pytest.log 2>&1
rc=$?
tail -n 40 gpurun_out/r2c34_pytest.log | cut -c1-300
echo "pytest rc=$rc"
