#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for smp in 0 1; do
for cfg in "1 1" "4 3"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --workload shipped --batch $1 --streams $2 --steps 50 --warmup 5 --no-cpu-baseline --no-exact --no-timer --mode sampler=$smp > gpurun_out/r2c20.json 2> gpurun_out/r2c20_err.log || { tail gpurun_out/r2c20_err.log; exit 1; }
  python - <<PY
import json
d=json.load(open('gpurun_out/r2c20.json')); print('shipped sampler=$smp $1x$2', round(d['value'],1), 'scenes/s', round(d['ms_per_step'],4), 'ms/step')
PY
done
done
for smp in 0 1; do
  timeout -k 10 300 python bench.py --workload v2xreal --batch 1 --streams 1 --steps 50 --warmup 5 --no-cpu-baseline --no-exact --no-timer --mode sampler=$smp > gpurun_out/r2c20.json 2> gpurun_out/r2c20_err.log || exit 1
  python - <<PY
import json
d=json.load(open('gpurun_out/r2c20.json')); print('v2xreal sampler=$smp 1x1', round(d['value'],1), 'scenes/s', round(d['ms_per_step'],4), 'ms/step')
PY
done
