#!/bin/bash
set -o pipefail
for i in 1 2; do
  for v in A B; do
    if [ $v = A ]; then export GENCOMM_HIP_LIB=$PWD/tools/ab/libgencomm_A.so; else unset GENCOMM_HIP_LIB; fi
    echo "== $v$i"; timeout -k 10 300 python tools/v2xvit_bench.py 2>&1 | grep -v amdgpu.ids | grep "v2xvit"
  done
done
