#!/bin/bash
# A/B on one box: 1x1 layers on the three-term f16-pipe kernel (default) against the exact-fp32 kernel (variant library built with -DH3_NO_1X1),
# on the training leg and on the large hot-path training step
set -o pipefail
O=gpurun_out; mkdir -p $O
for v in default no1x1 default no1x1; do
  if [ $v = no1x1 ]; then export GENCOMM_HIP_LIB=$PWD/gencomm_amd/libgencomm_no1x1.so; else unset GENCOMM_HIP_LIB; fi
  timeout -k 10 280 python bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline > $O/ab.json 2> $O/ab.err || { tail -n 5 $O/ab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$v: leg %.2f ms/step'%d['ms_per_step'])"
  timeout -k 10 280 python tools/train_bench.py --only large --batch 1 2>&1 | grep "train step" | sed "s/^/$v: /"
done 2>&1 | tee $O/r5_h3_1x1_ab.txt
