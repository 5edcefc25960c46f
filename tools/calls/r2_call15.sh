#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r2c15_prof -o tr -- python3 tools/train_bench.py > gpurun_out/r2c15_prof.log 2>&1 || { tail gpurun_out/r2c15_prof.log; exit 1; }
cat gpurun_out/r2c15_prof.log | tail -3
python tools/rocpd_stats.py gpurun_out/r2c15_prof/tr_results.db --by-grid > gpurun_out/r2c15_train_kernel_stats_by_grid.csv
python tools/rocpd_stats.py gpurun_out/r2c15_prof/tr_results.db > gpurun_out/r2c15_train_kernel_stats.csv
rm -rf gpurun_out/r2c15_prof
head -n 40 gpurun_out/r2c15_train_kernel_stats.csv | cut -c1-180
