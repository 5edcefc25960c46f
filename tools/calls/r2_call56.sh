#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_shell.py tests/test_gpu_backward.py tests/test_gpu_parity.py tests/test_msgext_oracle.py tests/test_gpu_configs.py -m gpu -q -x > gpurun_out/r2c56_pytest.log 2>&1
rc=$?
tail -n 4 gpurun_out/r2c56_pytest.log | cut -c1-250
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
for f in att v2xvit; do timeout -k 10 300 python tools/shell_bench.py --fusion $f 2>&1 | grep -v amdgpu.ids | tail -2; done
