#!/bin/bash
# concurrency experiments on one box: HW queue count, streams x batch
set -o pipefail
mkdir -p gpurun_out
run() { # label, env assignments..., -- bench args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py --no-cpu-baseline --no-exact --no-timer "$@" > gpurun_out/c38_$label.json 2> gpurun_out/c38_err.log || { tail gpurun_out/c38_err.log; return 1; }
  python - <<PY
import json
d=json.load(open('gpurun_out/c38_$label.json')); print('$label', round(d['value'],1))
PY
}
run base X=1 -- &&
run q8 GPU_MAX_HW_QUEUES=8 -- &&
run q8_s6 GPU_MAX_HW_QUEUES=8 -- --streams 6 --batch 2 &&
run q8_s4 GPU_MAX_HW_QUEUES=8 -- --streams 4 --batch 4 &&
run q2 GPU_MAX_HW_QUEUES=2 -- &&
run s2 X=1 -- --streams 2 --batch 4 &&
run s1 X=1 -- --streams 1 --batch 4 &&
run s1b8 X=1 -- --streams 1 --batch 8 &&
run base2 X=1 --
