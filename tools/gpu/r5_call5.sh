#!/bin/bash
# Round 5, call 5: ABI v9 (BatchNorm counter inside its kernel, pre-zeroed scratch from the zero pool, conv2d timer family) -- training
# tests, then the training leg eager and as one HIP graph per step.
set -o pipefail
O=gpurun_out; mkdir -p $O
echo "== tests"; timeout -k 10 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_ops.py tests/test_backbone.py tests/test_abi.py tests/test_shell.py tests/test_loss.py -m gpu -q -x > $O/r5c5_tests.log 2>&1; rc=$?; tail -n 4 $O/r5c5_tests.log; [ $rc -eq 0 ] || exit $rc
echo "== train leg eager"; timeout -k 10 400 python bench.py --workload train --steps 20 --warmup 3 > $O/r5c5_train_eager.json 2> $O/r5c5_train_eager.err || { tail -n 30 $O/r5c5_train_eager.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r5c5_train_eager.json'));r=d['roofline'];c=d['cpu_baseline'];print('eager: %.1f scenes/s, %.2f ms/step, host enqueue %.2f ms; roofline %s frac %.3f (%d launches, %.1f%% of the step); cpu %.2f scenes/s'%(d['value'],d['ms_per_step'],d['host_enqueue_ms_per_step'],r['kernel'][:24],r['frac'],r['launches'],100*r['share_of_step'],c['value']))"
echo "== train leg graph"; timeout -k 10 400 python bench.py --workload train --steps 20 --warmup 3 --graph 1 --no-cpu-baseline > $O/r5c5_train_graph.json 2> $O/r5c5_train_graph.err || { tail -n 30 $O/r5c5_train_graph.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r5c5_train_graph.json'));print('graph: %.1f scenes/s, %.2f ms/step, host enqueue %.2f ms, loss %s'%(d['value'],d['ms_per_step'],d['host_enqueue_ms_per_step'],d['loss']))"
echo "== rocprof train leg (launch census)"; export TMPDIR=/tmp; rm -rf $O/r5_prof_leg
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_prof_leg -o k -- python3 bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline > $O/r5_prof_leg.log 2>&1 || { tail -n 20 $O/r5_prof_leg.log; exit 1; }
find $O/r5_prof_leg -name "*kernel_stats.csv" -exec cp {} $O/r5_train_leg_kernel_stats.csv \;
rm -rf $O/r5_prof_leg
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5_train_leg_kernel_stats.csv')))
tot=sum(int(r['Calls']) for r in rows); tt=sum(float(r['TotalDurationNs']) for r in rows)
print('launches per step (8 steps incl. the timer step):', tot/8, 'kernel ms per step', tt/8e6)
rows.sort(key=lambda r:-int(r['Calls']))
for r in rows[:12]: print(int(r['Calls'])//8, '%7.1f us/step'%(float(r['TotalDurationNs'])/8e3), r['Name'][:90])
PY
echo done
