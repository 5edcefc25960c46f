#!/bin/bash
# last call of the round: complete GPU suite + smoke + the default bench line at HEAD
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r2final_pytest.log 2>&1
rc=$?
tail -n 4 gpurun_out/r2final_pytest.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -1
timeout -k 10 400 python bench.py > gpurun_out/r2final_bench.json 2> gpurun_out/r2final_err.log || { tail gpurun_out/r2final_err.log; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2final_bench.json')); r=d['roofline']
print('value', round(d['value'],1), 'frac', round(r['frac'],3), 'traffic', r['traffic'], 'latent ms', round(d['roofline_latent_step']['avg_launch_ms'],4), 'exact', round(d['exact_fp32_mode']['value'],1), 'cpu', round(d['cpu_baseline']['value'],4))
PY
