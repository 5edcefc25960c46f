#!/bin/bash
# conv8h instruction trims (cheap lane offsets, clamped tile loads, statistics init): conv / parity / config tests, then the default bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_conv8.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_properties.py tests/test_gpu_bf16.py -m gpu -q -x > gpurun_out/r2c37_pytest.log 2>&1
rc=$?
tail -n 12 gpurun_out/r2c37_pytest.log | cut -c1-240
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python bench.py --no-cpu-baseline --no-exact > gpurun_out/r2c37_bench.json 2> gpurun_out/r2c37_bench.err || { tail -n 20 gpurun_out/r2c37_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c37_bench.json'))
r=d['roofline']
print('value', round(d['value'],1), 'roofline frac', round(r['frac'],4), 'avg launch us', r.get('avg_launch_us') or r.get('avg_launch_ms'))
print({k:(round(v['achieved'],1) if isinstance(v,dict) and 'achieved' in v else v) for k,v in r.get('variants',{}).items()})
print('latent ms', d['roofline_latent_step']['avg_launch_ms'])
PY
