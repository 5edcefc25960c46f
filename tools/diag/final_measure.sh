#!/bin/bash
# end-of-round reference measurements: default bench line, rocprofv3 kernel stats of the metric workload with one stream, PMC traffic passes
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err || { tail -n 20 gpurun_out/r3_bench_default.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3_bench_default.json"))
print("default bench: %.1f scenes/s, ms_per_step %.2f, roofline frac %.3f (avg launch %.1f us, traffic %s), latent %.3f ms, exact %.1f, cpu %.4f" % (
    d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"]*1e3, d["roofline"]["traffic"], d["roofline_latent_step"]["avg_launch_ms"],
    d["exact_fp32_mode"]["value"], d["cpu_baseline"]["value"]))
PY
rm -rf gpurun_out/r3_prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof_final -o k -- python3 bench.py --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-exact --no-timer > gpurun_out/r3_prof_final.log 2>&1 || { tail -n 20 gpurun_out/r3_prof_final.log; exit 1; }
bash tools/pmc_pass.sh metric > gpurun_out/r3_pmc_pass.log 2>&1 || { tail -n 20 gpurun_out/r3_pmc_pass.log; exit 1; }
echo done
