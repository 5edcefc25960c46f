#!/bin/bash
# Round 5, call 4: full GPU suite on the diet library with the 16-pixel-wide partial-conv tiles (dc = 64 on the f16 pipe), then the
# shipped / v2xreal lines and the default line.
set -o pipefail
O=gpurun_out; mkdir -p $O
echo "== gpu tests"; timeout -k 10 700 python -m pytest tests -m gpu -q -x > $O/r5c4_gpu_tests.log 2>&1; rc=$?; tail -n 4 $O/r5c4_gpu_tests.log; [ $rc -eq 0 ] || exit $rc
for wl in shipped v2xreal; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --sustain 2 > $O/r5c4_$wl.json 2> $O/r5c4_$wl.err || { tail -n 20 $O/r5c4_$wl.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/r5c4_$wl.json'));print('$wl: %.1f scenes/s, latency one scene %.3f ms'%(d['value'],d['latency_ms_one_scene']));print({k[:30]:v for k,v in d['kernel_time_shares'].items()})"
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-exact --sustain 2 > $O/r5c4_default.json 2> $O/r5c4_default.err || { tail -n 20 $O/r5c4_default.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r5c4_default.json'));print('default: %.1f scenes/s, frac %.3f'%(d['value'],d['roofline']['frac']))"
