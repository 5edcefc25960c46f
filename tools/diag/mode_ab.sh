#!/bin/bash
# A/B of library modes on ONE box, alternating: bash tools/diag/mode_ab.sh <tag> <rounds> name=key=value[,key=value] ...
# (every variant is the shipped library; "base" = no --mode flag)
set -o pipefail
TAG=$1; ROUNDS=$2; shift 2
O=gpurun_out; mkdir -p $O
for r in $(seq 1 $ROUNDS); do
  for kv in base= "$@"; do
    name=${kv%%=*}; m=${kv#*=}
    flags=""
    if [ -n "$m" ]; then for one in ${m//,/ }; do flags="$flags --mode $one"; done; fi
    timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-exact --no-timer $flags > $O/${TAG}_${name}_$r.json 2> $O/${TAG}_${name}_$r.err || { tail -n 5 $O/${TAG}_${name}_$r.err; exit 1; }
    python -c "
import json,sys;d=json.load(open('$O/${TAG}_${name}_$r.json'));print('$name round $r: %.1f scenes/s (sustained %.1f)'%(d['value'], d['sustained']['value_this_rank'] if d['sustained'] else d['value']))"
  done
done
