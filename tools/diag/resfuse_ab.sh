#!/bin/bash
# A/B of the EMULATED ResnetBlock fusion (GENCOMM_MODE_RESFUSE_EMU, unet_host.h): alternating bench runs on one box.
set -e
for m in 0 1 0 1; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact --mode resfuse_emu=$m > gpurun_out/r3_resfuse_$m.json 2> gpurun_out/r3_resfuse.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3_resfuse_$m.json"))
r=d["roofline"]
v={x["variant"][:26]:round(x["avg_launch_ms"]*1e3,1) for x in r["variants"]}
print("resfuse_emu=$m  %.1f scenes/s  ms_per_step %.2f  conv8h family avg %.1f us  variants(us) %s" % (d["value"], d["ms_per_step"], r["avg_launch_ms"]*1e3, v))
PY
done
