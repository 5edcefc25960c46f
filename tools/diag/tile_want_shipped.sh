#!/bin/bash
# shipped / v2xreal shapes: minimum workgroup count from which the f16-pipe tile kernels replace the exact-fp32 small-map kernel (GENCOMM_MODE_TILE_WANT)
set -o pipefail
O=gpurun_out; mkdir -p $O
for wl in shipped v2xreal; do
  for tw in 0 96 64 32 1 0; do
    timeout -k 10 200 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-exact --sustain 2 --mode tile_want=$tw > $O/tw.json 2> $O/tw.err || { tail -n 5 $O/tw.err; exit 1; }
    python -c "
import json; d=json.load(open('gpurun_out/tw.json')); print('$wl tile_want=$tw: %.1f scenes/s (sustained %.1f), latency %.3f ms'%(d['value'], d['sustained']['value_this_rank'], d['latency_ms_one_scene']))"
  done
done 2>&1 | tee $O/r5_tile_want_shipped.txt
