#!/bin/bash
O=gpurun_out; mkdir -p $O
for m in 0 1; do
  timeout -k 10 200 python bench.py --workload shipped --no-cpu-baseline --no-exact --no-timer --mode dataflow=$m > $O/r4c16_df$m.json 2> $O/r4c16_df$m.err || { tail -5 $O/r4c16_df$m.err; }
  python -c "
import json;d=json.load(open('gpurun_out/r4c16_df$m.json'));print('dataflow=$m shipped: %.0f scenes/s'%d['value'])"
  timeout -k 10 200 python bench.py --workload shipped --no-cpu-baseline --no-exact --mode dataflow=$m --streams 1 --batch 1 --steps 200 > $O/r4c16_df${m}_b1.json 2> $O/r4c16_df${m}_b1.err || { tail -5 $O/r4c16_df${m}_b1.err; }
  python -c "
import json;d=json.load(open('gpurun_out/r4c16_df${m}_b1.json'));print('dataflow=$m shipped 1 stream x 1 scene: %.0f scenes/s, %.3f ms/step'%(d['value'], d['ms_per_step']))"
done
