#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
rm -rf $O/r4c14_prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4c14_prof -o k -- python3 bench.py --workload train --steps 5 --warmup 2 > $O/r4c14_prof.log 2>&1 || { tail -n 20 $O/r4c14_prof.log; exit 1; }
find $O/r4c14_prof -name "*kernel_stats.csv" -exec cp {} $O/r4c14_leg_kernel_stats.csv \;
python tools/trace_by_grid.py $(find $O/r4c14_prof -name "*kernel_trace.csv" | head -1) conv2d_igemm wgrad3x3 bn2d dcn > $O/r4c14_leg_by_grid.csv
rm -rf $O/r4c14_prof
head -n 30 $O/r4c14_leg_by_grid.csv | cut -c1-170
