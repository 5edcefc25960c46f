#!/bin/bash
# per-family kernel times of two libraries on one box (HIP-event family pass of bench.py)
set -o pipefail
O=gpurun_out; mkdir -p $O
for kv in new= "$@"; do
  name=${kv%%=*}; path=${kv#*=}
  if [ -n "$path" ]; then export GENCOMM_HIP_LIB=$PWD/$path; else unset GENCOMM_HIP_LIB; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact --sustain 1 > $O/r5c3_$name.json 2> $O/r5c3_$name.err || { tail -n 5 $O/r5c3_$name.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r5c3_$name.json"))
r=d["roofline"]
print("$name: %.1f scenes/s; conv8h family avg %.2f us frac %.3f; latent %.1f us; "%(d["value"], r["avg_launch_ms"]*1e3, r["frac"], d["roofline_latent_step"]["avg_launch_ms"]*1e3) + "; ".join("%s %.1f us"%(v["variant"][:22], v["avg_launch_ms"]*1e3) for v in r["variants"]))
print("   shares:", {k[:28]:v for k,v in list(d["kernel_time_shares"].items())[:12]})
PY
done
