#!/bin/bash
# Round 5, call 6: the persistent form of the 8-channel f16-pipe kernels (GENCOMM_MODE_PERSIST) -- equality test, then per-variant timing
set -o pipefail
O=gpurun_out; mkdir -p $O
echo "== tests"; timeout -k 10 400 python -m pytest tests/test_gpu_conv8.py -m gpu -q -x -k "persistent" > $O/r5c6_tests.log 2>&1; rc=$?; tail -n 6 $O/r5c6_tests.log; [ $rc -eq 0 ] || exit $rc
for m in 0 31 2 8 4 1 16 10 0 31; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact --sustain 1 --mode persist=$m > $O/r5c6_p$m.json 2> $O/r5c6_p$m.err || { tail -n 5 $O/r5c6_p$m.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r5c6_p$m.json"))
r=d["roofline"]
print("persist=$m: %.1f scenes/s; conv8h family avg %.2f us frac %.3f; "%(d["value"], r["avg_launch_ms"]*1e3, r["frac"]) + "; ".join("%s %.1f"%(v["variant"][:18], v["avg_launch_ms"]*1e3) for v in r["variants"]))
PY
done 2>&1 | tee $O/r5_persist_ab.txt
