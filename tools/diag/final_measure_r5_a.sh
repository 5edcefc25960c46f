#!/bin/bash
# End-of-round reference measurements, part A (round 5, one box per part; parts B and C: final_measure_r5_b.sh / _c.sh): bench lines of the four
# inference workloads, the self-launched 4-rank share-device rehearsal, the bf16 denoise mode.  Outputs under gpurun_out/r5_*.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
step() { echo "== $1"; }
step "bench default"; timeout -k 10 400 python bench.py > $O/r5_bench_default_a.json 2> $O/r5_bench_default_a.err || { tail -n 20 $O/r5_bench_default_a.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_bench_default_a.json"))
print("default: %.1f scenes/s, ms/step %.2f, latency one scene %.2f ms, sustained %s, roofline frac %.3f (avg launch %.1f us, traffic %s), latent %.3f ms, exact %.1f, cpu %.4f (%d threads)" % (
    d["value"], d["ms_per_step"], d["latency_ms_one_scene"], d["sustained"] and round(d["sustained"]["value_this_rank"],1), d["roofline"]["frac"], d["roofline"]["avg_launch_ms"]*1e3,
    d["roofline"]["traffic"], d["roofline_latent_step"]["avg_launch_ms"], d["exact_fp32_mode"]["value"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
PY
for wl in cfg2 shipped v2xreal; do
  step "bench $wl"; timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --sustain 3 > $O/r5_bench_$wl.json 2> $O/r5_bench_$wl.err || { tail -n 20 $O/r5_bench_$wl.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/r5_bench_$wl.json'));print('$wl: %.1f scenes/s (sustained %.1f), latency one scene %.3f ms'%(d['value'],d['sustained']['value_this_rank'],d['latency_ms_one_scene']))"
done
step "bench --gpus 4 --share-device (self-launched)"; timeout -k 10 300 python bench.py --gpus 4 --share-device --steps 20 --warmup 3 --no-exact --no-timer > $O/r5_bench_share4.json 2> $O/r5_bench_share4.err || { tail -n 20 $O/r5_bench_share4.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r5_bench_share4.json'));print('share4: %.1f scenes/s over %d ranks, %d scenes, group %s'%(d['value'],d['n_gpus'],d['total_scenes'],d['config']['process_group']))"
for wl in metric shipped; do
  step "bf16 $wl"; timeout -k 10 300 python bench.py --workload $wl --mode arith=2 --no-cpu-baseline --no-exact --sustain 2 > $O/r5_bench_bf16_$wl.json 2> $O/r5_bench_bf16_$wl.err || { tail -n 20 $O/r5_bench_bf16_$wl.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/r5_bench_bf16_$wl.json'));print('bf16 $wl: %.1f scenes/s'%d['value'])"
done
echo done
