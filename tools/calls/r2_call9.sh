#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for cfg in "4 3" "4 4" "2 4" "2 6" "8 2" "6 3" "3 4" "8 3"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --steps $((80 / $1)) --warmup 3 --no-cpu-baseline --no-exact --no-timer --batch $1 --streams $2 > gpurun_out/r2c9_b$1_s$2.json 2>/dev/null || { echo "fail $cfg"; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/r2c9_b$1_s$2.json')); print('batch $1 streams $2:', round(d['value'],1), 'scenes/s')"
done
