#!/bin/bash
# Round 5, call 8: device idle accounting of the training leg (kernel trace of 8 steps)
set -o pipefail
O=gpurun_out; mkdir -p $O; rm -rf $O/prof_leg
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof_leg -o k -- python3 bench.py --workload train --steps 8 --warmup 3 --no-cpu-baseline --no-timer > $O/prof_leg.log 2>&1 || { tail -n 20 $O/prof_leg.log; exit 1; }
f=$(find $O/prof_leg -name "*kernel_trace.csv" | head -n 1)
python tools/trace_idle.py $f 0.5 | tee $O/r5_train_idle.txt
grep -o '"ms_per_step": [0-9.]*' $O/prof_leg.log | head -n 1
rm -rf $O/prof_leg
