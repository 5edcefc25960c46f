#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r2c18_prof -o sh -- python3 bench.py --workload shipped --batch 1 --streams 1 --steps 50 --warmup 5 --no-cpu-baseline --no-exact --no-timer > gpurun_out/r2c18_prof.log 2>&1 || { tail gpurun_out/r2c18_prof.log; exit 1; }
tail -n 2 gpurun_out/r2c18_prof.log | cut -c1-300
python tools/rocpd_stats.py gpurun_out/r2c18_prof/sh_results.db --by-grid > gpurun_out/r2c18_shipped_1x1_kernel_stats_by_grid.csv
rm -rf gpurun_out/r2c18_prof
head -n 40 gpurun_out/r2c18_shipped_1x1_kernel_stats_by_grid.csv | cut -c1-170
