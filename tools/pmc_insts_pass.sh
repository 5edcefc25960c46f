#!/bin/bash
# Dynamic instruction mix per wave of the benchmark's kernels: one SQ PMC pass (8 slots).
set -o pipefail
export R=${ROUND:-r5}   # prefix of the output files (profiles/<round>_pmc_*.json)
WL=${1:-metric}
export PMC_WORKLOAD=$WL
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/${R}_pmc_INSTS
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT --output-format csv -d gpurun_out/${R}_pmc_INSTS -o pmc -- python3 bench.py --workload $WL --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-exact --no-timer > gpurun_out/${R}_pmc_INSTS.log 2>&1 || { tail -n 20 gpurun_out/${R}_pmc_INSTS.log; exit 1; }
python - <<'PY'
import csv, collections, json, os, re
rows = csv.DictReader(open('gpurun_out/' + os.environ['R'] + '_pmc_INSTS/pmc_counter_collection.csv'))
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
    w = m.get('SQ_WAVES', 0.0)
    if w <= 0:
        continue
    out[k] = {"launches": int(d['SQ_WAVES'][1]), "waves_per_launch": round(w, 1),
              "per_wave": {c[len('SQ_INSTS_'):].lower(): round(v / w, 1) for c, v in m.items() if c != 'SQ_WAVES'}}
res = {"_about": "dynamic instructions per wave (one SQ PMC pass, tools/pmc_insts_pass.sh): VALU includes MFMA, transcendental and conversion instructions",
       "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]['waves_per_launch'] * kv[1]['launches'] * kv[1]['per_wave'].get('valu', 0)))}
import os, sys
sys.path.insert(0, os.getcwd())
from gencomm_amd import _lib
res['library_src'] = _lib.library_src_hash()   # the kernels these counters were taken on (bench.py ignores a file from another library)
res['workload'] = os.environ.get('PMC_WORKLOAD', 'metric')
json.dump(res, open('gpurun_out/' + os.environ['R'] + '_pmc_insts.json', 'w'), indent=1)
for k, v in list(res['kernels'].items())[:10]:
    print(k[:60].ljust(60), v['launches'], v['waves_per_launch'], v['per_wave'])
PY
