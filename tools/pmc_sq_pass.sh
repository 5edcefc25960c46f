#!/bin/bash
# Where the waves of the benchmark's kernels spend their cycles: one rocprofv3 --pmc pass of the SQ block (8 slots on gfx950,
# MI355X_MICROARCH.md "rocprofv3 PMC slots"): SQ_WAIT_ANY (parked on s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall) +
# SQ_ACTIVE_INST_ANY (issuing) ~ SQ_WAVE_CYCLES; SQ_VALU_MFMA_BUSY_CYCLES; LDS conflict / active cycles.
#   bash tools/pmc_sq_pass.sh [workload]   (on the GPU box; output gpurun_out/${R}_pmc_sq.json)
set -o pipefail
export R=${ROUND:-r5}   # prefix of the output files (profiles/<round>_pmc_*.json)
WL=${1:-metric}
export PMC_WORKLOAD=$WL
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/${R}_pmc_SQ
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/${R}_pmc_SQ -o pmc -- python3 bench.py --workload $WL --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-exact --no-timer > gpurun_out/${R}_pmc_SQ.log 2>&1 || { tail -n 20 gpurun_out/${R}_pmc_SQ.log; exit 1; }
python - <<'PY'
import csv, collections, json, os, re
rows = csv.DictReader(open('gpurun_out/' + os.environ['R'] + '_pmc_SQ/pmc_counter_collection.csv'))
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in rows:
    k = re.sub(r'^void ', '', r['Kernel_Name'])
    if not k.startswith('gc::'):
        continue
    a = acc[k][r['Counter_Name']]
    a[0] += float(r['Counter_Value']); a[1] += 1
out = {}
for k, d in acc.items():
    m = {c: v[0] / v[1] for c, v in d.items()}
    wc = m.get('SQ_WAVE_CYCLES', 0.0)
    if wc <= 0:
        continue
    out[k] = {"launches": int(d['SQ_WAVE_CYCLES'][1]),
              "wait_any_frac": round(m.get('SQ_WAIT_ANY', 0) / wc, 3),
              "wait_inst_frac": round(m.get('SQ_WAIT_INST_ANY', 0) / wc, 3),
              "active_inst_frac": round(m.get('SQ_ACTIVE_INST_ANY', 0) / wc, 3),
              "mfma_busy_over_busy_cycles": round(m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(m.get('SQ_BUSY_CYCLES', 1), 1), 3),
              "lds_conflict_over_lds_active": round(m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1), 3),
              "raw_means": {c: round(v, 1) for c, v in m.items()}}
res = {"_about": "per-kernel means of one SQ PMC pass (tools/pmc_sq_pass.sh): fractions of SQ_WAVE_CYCLES (quad-cycle units) the waves spend "
                 "parked on s_waitcnt / barriers (wait_any), stalled at issue (wait_inst), issuing (active_inst); MFMA busy cycles over SQ busy cycles; "
                 "LDS bank-conflict cycles over LDS active cycles", "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]['raw_means'].get('SQ_WAVE_CYCLES', 0) * kv[1]['launches']))}
import os, sys
sys.path.insert(0, os.getcwd())
from gencomm_amd import _lib
res['library_src'] = _lib.library_src_hash()   # the kernels these counters were taken on (bench.py ignores a file from another library)
res['workload'] = os.environ.get('PMC_WORKLOAD', 'metric')
json.dump(res, open('gpurun_out/' + os.environ['R'] + '_pmc_sq.json', 'w'), indent=1)
for k, v in list(res['kernels'].items())[:14]:
    print(k[:64].ljust(64), v['launches'], 'wait', v['wait_any_frac'], 'stall', v['wait_inst_frac'], 'active', v['active_inst_frac'], 'mfma', v['mfma_busy_over_busy_cycles'], 'ldsconf', v['lds_conflict_over_lds_active'])
PY
