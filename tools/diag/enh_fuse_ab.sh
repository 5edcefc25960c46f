set -e
for m in 2 1 2 1; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact --mode enh_fuse=$m > gpurun_out/r3_d2_fuse$m.json 2> gpurun_out/r3_d2.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r3_d2_fuse$m.json"))
sh=d["kernel_time_shares"]
print("enh_fuse=$m value %.1f"%d["value"], {k[:28]:v for k,v in sh.items() if "enh" in k or "gemm" in k or "warp" in k})
PY
done
