#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r2c14_prof -o v2x -- python3 tools/v2xvit_bench.py > gpurun_out/r2c14_prof.log 2>&1 || { tail gpurun_out/r2c14_prof.log; exit 1; }
python tools/rocpd_stats.py gpurun_out/r2c14_prof/v2x_results.db --by-grid > gpurun_out/r2c14_v2xvit_kernel_stats.csv
rm -rf gpurun_out/r2c14_prof
head -n 30 gpurun_out/r2c14_v2xvit_kernel_stats.csv | cut -c1-200
