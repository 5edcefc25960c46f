#!/bin/bash
# Round 5, call 10: the training leg with every launch on ONE stream (GENCOMM_MODE_BWD_STREAMS = 0): isolated kernel durations
set -o pipefail
O=gpurun_out; mkdir -p $O
timeout -k 10 280 python bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline --mode bwd_streams=0 > $O/r5_train_1s.json 2> $O/r5_train_1s.err || { tail -n 20 $O/r5_train_1s.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r5_train_1s.json')); print('one stream: %.2f ms/step, host %.2f ms'%(d['ms_per_step'], d['host_enqueue_ms_per_step']))"
rm -rf $O/prof_leg
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_leg -o k -- python3 bench.py --workload train --steps 5 --warmup 3 --no-cpu-baseline --no-timer --mode bwd_streams=0 > $O/prof_leg.log 2>&1 || { tail -n 20 $O/prof_leg.log; exit 1; }
find $O/prof_leg -name "*kernel_stats.csv" -exec cp {} $O/r5_train_1s_kernel_stats.csv \;
rm -rf $O/prof_leg
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/r5_train_1s_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step %.2f ms"%(tot/9/1e6))
for r in rows[:45]: print("%5.1f%% %6.1f/step %8.1f us  %6.3f ms/step  %s"%(100*float(r["TotalDurationNs"])/tot, int(r["Calls"])/9, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/9e6, r["Name"][:90]))
PY
