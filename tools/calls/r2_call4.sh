#!/bin/bash
# round 2, GPU call 4: new bench.py (family roofline, exact mode at full steps, CPU baseline with CI), PMC passes
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > gpurun_out/r2c4_bench.json 2> gpurun_out/r2c4_bench.err || { tail -n 30 gpurun_out/r2c4_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c4_bench.json'))
r=d['roofline']; l=d['roofline_latent_step']; e=d['exact_fp32_mode']; c=d['cpu_baseline']
print('value', d['value'], 'ms/step', d['ms_per_step'])
print('roofline', r['achieved'], r['frac'], r['avg_launch_ms'], r['share_of_kernel_time'], r['traffic'])
for v in r['variants']: print('   ', v)
print('latent', l['avg_launch_ms'], l['frac'], l['share_of_kernel_time'])
print('exact', e['value'], e.get('roofline',{}).get('achieved'))
print('cpu', c['value'], c['ci95'], c['cores'])
print(d['kernel_time_shares'])
PY
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact --workload shipped --batch 1 --streams 1 > gpurun_out/r2c4_bench_shipped_1x1.json 2>> gpurun_out/r2c4_bench.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r2c4_bench_shipped_1x1.json')); print('shipped 1x1', d['value'], d['ms_per_step'])"
timeout -k 10 600 bash tools/pmc_pass.sh metric || exit 1
