#!/bin/bash
# generator trims: noise-related tests, then the default bench line (latent step duration is in roofline_latent_step)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_properties.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_conv8.py -m gpu -q -x > gpurun_out/r2c36_pytest.log 2>&1
rc=$?
tail -n 12 gpurun_out/r2c36_pytest.log | cut -c1-240
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python bench.py --no-cpu-baseline --no-exact > gpurun_out/r2c36_bench.json 2> gpurun_out/r2c36_bench.err || { tail -n 20 gpurun_out/r2c36_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c36_bench.json'))
print('value', round(d['value'],1), 'roofline', d['roofline']['frac'], 'latent', d.get('roofline_latent_step'))
PY
