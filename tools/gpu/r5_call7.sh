#!/bin/bash
# Round 5, call 7: the training leg on the three-term general convolution (conv2d_h3_kernel) -- bench line + rocprofv3 kernel statistics
set -o pipefail
O=gpurun_out; mkdir -p $O
T=${1:-h3}
timeout -k 10 280 python bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline > $O/r5_train_$T.json 2> $O/r5_train_$T.err || { tail -n 20 $O/r5_train_$T.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r5_train_$T.json")); r=d["roofline"]
print("train leg: %.2f ms/step, host %.2f ms; family: %d launches, avg %.1f us, %.1f TFLOP/s fp32-equivalent, share %.2f"%(d["ms_per_step"], d["host_enqueue_ms_per_step"], r["launches"], r["avg_launch_ms"]*1e3, r["achieved"], r["share_of_step"]))
PY
rm -rf $O/prof_leg
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_leg -o k -- python3 bench.py --workload train --steps 5 --warmup 3 --no-cpu-baseline --no-timer > $O/prof_leg.log 2>&1 || { tail -n 20 $O/prof_leg.log; exit 1; }
find $O/prof_leg -name "*kernel_stats.csv" -exec cp {} $O/r5_train_${T}_kernel_stats.csv \;
rm -rf $O/prof_leg
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/r5_train_${T}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step %.2f ms"%(tot/8/1e6))
for r in rows[:28]: print("%5.1f%% %6.1f/step %8.1f us  %s"%(100*float(r["TotalDurationNs"])/tot, int(r["Calls"])/8, float(r["AverageNs"])/1e3, r["Name"][:100]))
PY
