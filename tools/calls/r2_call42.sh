#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_v2xvit.py tests/test_where2comm.py tests/test_backbone.py tests/test_late.py -m gpu -q -x > gpurun_out/r2c42_pytest.log 2>&1
rc=$?
tail -n 5 gpurun_out/r2c42_pytest.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/calls/r2_call40.sh > /dev/null 2>&1
grep -v amdgpu.ids gpurun_out/c40_prof.log | tail -8
grep "conv1x1\|conv2d_igemm_kernel<1" gpurun_out/c40_kernel_stats_by_grid.csv | cut -c1-150 | head -12
