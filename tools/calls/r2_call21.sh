#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_v2xvit.py tests/test_shell.py tests/test_backbone.py -m gpu -q -x > gpurun_out/r2c21_pytest.log 2>&1
rc=$?
tail -n 12 gpurun_out/r2c21_pytest.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/v2xvit_bench.py > gpurun_out/r2c21_v2xvit.log 2>&1 || { tail gpurun_out/r2c21_v2xvit.log; exit 1; }
cat gpurun_out/r2c21_v2xvit.log
