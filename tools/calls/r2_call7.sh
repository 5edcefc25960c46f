#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q -s > gpurun_out/r2c7_pytest.log 2>&1
rc=$?
grep -n "bf16 denoise\|bf16 vs\|passed\|failed" gpurun_out/r2c7_pytest.log | cut -c1-220
tail -n 25 gpurun_out/r2c7_pytest.log
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit 0; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact --mode arith=2 > gpurun_out/r2c7_bench_bf16.json 2> gpurun_out/r2c7_bench_bf16.err || { tail -n 20 gpurun_out/r2c7_bench_bf16.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c7_bench_bf16.json'))
r=d['roofline']
print('bf16 value', d['value'], d['dtype'], 'conv8 family', r['achieved'], r['avg_launch_ms'], r['share_of_kernel_time'])
for v in r['variants']: print('   ', v)
print(d['roofline_latent_step']['avg_launch_ms'])
PY
