#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/c58_prof -o c58 -- python3 tools/train_bench.py > gpurun_out/c58_prof.log 2>&1 || { tail -n 20 gpurun_out/c58_prof.log; exit 1; }
python tools/rocpd_stats.py gpurun_out/c58_prof/c58_results.db > gpurun_out/c58_kernel_stats.csv
python tools/rocpd_stats.py gpurun_out/c58_prof/c58_results.db --by-grid > gpurun_out/c58_kernel_stats_by_grid.csv
rm -rf gpurun_out/c58_prof
head -n 26 gpurun_out/c58_kernel_stats.csv | cut -c1-150
