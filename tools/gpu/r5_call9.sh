#!/bin/bash
# Round 5, call 9: per-(kernel, grid) statistics of the training leg for the kernels named on the command line
set -o pipefail
O=gpurun_out; mkdir -p $O; rm -rf $O/prof_leg
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof_leg -o k -- python3 bench.py --workload train --steps 5 --warmup 3 --no-cpu-baseline --no-timer > $O/prof_leg.log 2>&1 || { tail -n 20 $O/prof_leg.log; exit 1; }
f=$(find $O/prof_leg -name "*kernel_trace.csv" | head -n 1)
python tools/trace_by_grid.py $f "$@" > $O/r5_train_by_grid.csv
head -n 60 $O/r5_train_by_grid.csv | cut -c1-200
rm -rf $O/prof_leg
