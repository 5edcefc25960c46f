#!/bin/bash
# A/B on ONE box: tools/ab/libgencomm_A.so (reference build) against the in-tree library, alternating runs of the default bench line
set -o pipefail
mkdir -p gpurun_out
for i in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then export GENCOMM_HIP_LIB=$PWD/tools/ab/libgencomm_A.so; else unset GENCOMM_HIP_LIB; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-exact > gpurun_out/ab_${v}${i}.json 2> gpurun_out/ab_err.log || { tail gpurun_out/ab_err.log; exit 1; }
    python - <<PY
import json
d=json.load(open('gpurun_out/ab_${v}${i}.json')); r=d['roofline']
print('${v}${i}', round(d['value'],1), 'family us', round(r['avg_launch_ms']*1000,2), [round(x['avg_launch_ms']*1000,2) for x in r['variants']], 'latent', round(d['roofline_latent_step']['avg_launch_ms'],4))
PY
  done
done
