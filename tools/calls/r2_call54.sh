#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_shell.py tests/test_gpu_backward.py tests/test_pillars.py tests/test_where2comm.py tests/test_v2xvit.py tests/test_gpu_parity.py tests/test_late.py -m gpu -q -x > gpurun_out/r2c54_pytest.log 2>&1
rc=$?
tail -n 4 gpurun_out/r2c54_pytest.log | cut -c1-250
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
for f in att v2xvit where2comm; do timeout -k 10 300 python tools/shell_bench.py --fusion $f 2>&1 | grep -v amdgpu.ids | tail -2; done | tee gpurun_out/r2c54_shell.log
timeout -k 10 300 python tools/shell_bench.py --agents 5 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a gpurun_out/r2c54_shell.log
