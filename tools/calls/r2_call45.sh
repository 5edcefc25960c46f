#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/c45_prof -o c45 -- python3 tools/backbone_bench.py --n 2 > gpurun_out/c45_prof.log 2>&1 || { tail -n 20 gpurun_out/c45_prof.log; exit 1; }
python tools/rocpd_stats.py gpurun_out/c45_prof/c45_results.db --by-grid > gpurun_out/c45_kernel_stats_by_grid.csv
rm -rf gpurun_out/c45_prof
head -n 24 gpurun_out/c45_kernel_stats_by_grid.csv | cut -c1-170
