#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/c55_prof -o c55 -- python3 tools/shell_bench.py --iters 20 > gpurun_out/c55_prof.log 2>&1 || { tail -n 20 gpurun_out/c55_prof.log; exit 1; }
python tools/rocpd_stats.py gpurun_out/c55_prof/c55_results.db > gpurun_out/c55_kernel_stats.csv
rm -rf gpurun_out/c55_prof
head -n 45 gpurun_out/c55_kernel_stats.csv | cut -c1-150
