#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_conv8.py tests/test_gpu_properties.py tests/test_gpu_bf16.py -m gpu -q -x > gpurun_out/r2c25_pytest.log 2>&1
rc=$?
tail -n 6 gpurun_out/r2c25_pytest.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-exact > gpurun_out/r2c25_bench.json 2> gpurun_out/r2c25_err.log || { tail gpurun_out/r2c25_err.log; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c25_bench.json')); r=d['roofline']
print('value', round(d['value'],2), 'family avg us', round(1e3*r['avg_launch_ms'],2), 'frac', round(r['frac'],3), [(v['variant'][:12], round(1e3*v['avg_launch_ms'],1)) for v in r['variants']], 'latent', round(1e3*d['roofline_latent_step']['avg_launch_ms'],1))
PY
bash tools/pmc_sq_pass.sh metric | head -8
