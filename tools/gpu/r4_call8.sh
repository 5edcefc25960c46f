#!/bin/bash
set -o pipefail
O=gpurun_out; mkdir -p $O
for v in base wgd1 wgd2 wgd4; do
  if [ $v = base ]; then unset GENCOMM_HIP_LIB; else export GENCOMM_HIP_LIB=$PWD/gencomm_amd/_build/variants/lib_$v.so; fi
  echo "== $v"; timeout -k 10 120 python tools/diag/wgrad_time.py 2>&1 | grep wgrad
done
