#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_v2xvit.py tests/test_where2comm.py -m gpu -q -x > gpurun_out/r2c67_pytest.log 2>&1
rc=$?
tail -n 6 gpurun_out/r2c67_pytest.log | cut -c1-250
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/v2xvit_bench.py 2>&1 | grep -v amdgpu.ids
