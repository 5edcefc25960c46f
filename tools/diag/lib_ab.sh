#!/bin/bash
# A/B of diagnostic library builds on ONE box, alternating: bash tools/diag/lib_ab.sh <tag> <rounds> name=path.so [name=path.so ...]
# (a build: GENCOMM_HIP_LIB=<path> GENCOMM_EXTRA_FLAGS="-D..." python -c "from gencomm_amd import _lib; _lib.build(force=True)")
set -o pipefail
TAG=$1; ROUNDS=$2; shift 2
O=gpurun_out; mkdir -p $O
for r in $(seq 1 $ROUNDS); do
  for kv in base= "$@"; do
    name=${kv%%=*}; path=${kv#*=}
    if [ -n "$path" ]; then export GENCOMM_HIP_LIB=$PWD/$path; else unset GENCOMM_HIP_LIB; fi
    timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-exact --no-timer > $O/${TAG}_${name}_$r.json 2> $O/${TAG}_${name}_$r.err || { tail -n 5 $O/${TAG}_${name}_$r.err; exit 1; }
    python -c "
import json,sys;d=json.load(open('$O/${TAG}_${name}_$r.json'));print('$name round $r: %.1f scenes/s (sustained %.1f)'%(d['value'], d['sustained']['value_this_rank'] if d['sustained'] else d['value']))"
  done
done
