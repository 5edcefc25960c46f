#!/bin/bash
# Steady-state kernel launches per step of the training leg: rocprofv3 --kernel-trace --stats of bench.py --workload train at 4 and at 9 timed
# steps; the difference of the per-kernel call counts / 5 leaves out everything that runs once (optimizer state, first-use caches, warm-up).
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
for k in 4 9; do
  rm -rf $O/legp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/legp -o k -- python3 bench.py --workload train --steps $k --warmup 2 --no-cpu-baseline --no-timer > $O/legp_$k.log 2>&1 || { tail -n 20 $O/legp_$k.log; exit 1; }
  find $O/legp -name "*kernel_stats.csv" -exec cp {} $O/legp_kernel_stats_$k.csv \;
  rm -rf $O/legp
done
python - <<'PY'
import csv
def load(k):
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(f'gpurun_out/legp_kernel_stats_{k}.csv'))}
a, b = load(4), load(9)
rows = []
for name, (cb, tb) in b.items():
    ca, ta = a.get(name, (0, 0.0))
    if cb > ca:
        rows.append(((cb - ca) / 5.0, (tb - ta) / 5.0 / 1e3, name))
rows.sort(reverse=True)
print('steady state: %.0f launches per step, %.2f ms of kernel time per step, %d kernel names' % (sum(r[0] for r in rows), sum(r[1] for r in rows) / 1e3, len(rows)))
lib = sum(r[0] for r in rows if 'gc::' in r[2]); print('library kernels %.0f, framework / runtime kernels %.0f' % (lib, sum(r[0] for r in rows) - lib))
for c, t, name in rows[:60]:
    print('%7.1f per step %8.1f us per step  %s' % (c, t, name[:130]))
PY
