#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_second.py tests/test_shell.py -m gpu -q -x > gpurun_out/r2c51_pytest.log 2>&1
rc=$?
tail -n 12 gpurun_out/r2c51_pytest.log | cut -c1-250
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/second_bench.py 2>&1 | grep -v amdgpu.ids | grep "SECOND forward"
