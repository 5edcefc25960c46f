#!/bin/bash
# kernel trace of the V2X-ViT / Where2comm forward bench
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/c40_prof -o c40 -- python3 tools/v2xvit_bench.py > gpurun_out/c40_prof.log 2>&1 || { tail -n 20 gpurun_out/c40_prof.log; exit 1; }
python tools/rocpd_stats.py gpurun_out/c40_prof/c40_results.db --by-grid > gpurun_out/c40_kernel_stats_by_grid.csv
python tools/rocpd_stats.py gpurun_out/c40_prof/c40_results.db > gpurun_out/c40_kernel_stats.csv
rm -rf gpurun_out/c40_prof
head -n 30 gpurun_out/c40_kernel_stats.csv | cut -c1-200
