#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_msgext_oracle.py tests/test_gpu_backward.py tests/test_shell.py -m gpu -q -x -k "message or msgext or extractor or shell" > gpurun_out/r2c53_pytest.log 2>&1
rc=$?
tail -n 4 gpurun_out/r2c53_pytest.log | cut -c1-250
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/shell_bench.py 2>&1 | grep -v amdgpu.ids | tail -2
timeout -k 10 300 python tools/msgext_bench.py 2>&1 | grep -v amdgpu.ids | tail -6
