#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r2c19_pytest.log 2>&1
rc=$?
tail -n 12 gpurun_out/r2c19_pytest.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
for cfg in "1 1" "4 3"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --workload shipped --batch $1 --streams $2 --steps 50 --warmup 5 --no-cpu-baseline --no-exact --no-timer > gpurun_out/r2c19_shipped_$1x$2.json 2> gpurun_out/r2c19_err.log || { tail gpurun_out/r2c19_err.log; exit 1; }
  python - <<PY
import json
d=json.load(open('gpurun_out/r2c19_shipped_$1x$2.json')); print('shipped $1x$2', round(d['value'],1), 'scenes/s', d['ms_per_step'], 'ms/step')
PY
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-exact > gpurun_out/r2c19_bench.json 2> gpurun_out/r2c19_err.log || { tail gpurun_out/r2c19_err.log; exit 1; }
cut -c1-120 gpurun_out/r2c19_bench.json
