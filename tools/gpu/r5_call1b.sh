#!/bin/bash
# Round 5, call 1b (the full suite ran in call 1: 319 passed + the two new robustness tests, whose bounds were then set from their printed margins): full GPU suite on the round's first library, baseline bench line (12 s sustained region, MFMA-pipe rooflines), share-device
# rehearsals at 4 ranks (the pool's process guard allows at most 6 processes on the card: 8 ranks on ONE device cannot be run here),
# bf16 denoise mode on the metric and shipped workloads.
set -o pipefail
O=gpurun_out; mkdir -p $O
step() { echo "== $1"; }
step "robustness lines"; timeout -k 10 200 python -m pytest tests/test_gpu_robustness.py -m gpu -q -s 2>&1 | grep -E "^robust|passed|failed" | tee $O/r5_robustness.txt
step "bench default"; timeout -k 10 400 python bench.py > $O/r5_bench_default_pre.json 2> $O/r5_bench_default_pre.err || { tail -n 20 $O/r5_bench_default_pre.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_bench_default_pre.json"))
print("default: %.1f scenes/s, ms/step %.2f, latency %.2f ms, sustained %s s -> %s, roofline frac %.3f (avg %.1f us), latent frac(f16 pipe) %.3f fp32eq %.3f, enh front %s, exact %.1f, cpu %.4f" % (
    d["value"], d["ms_per_step"], d["latency_ms_one_scene"], d["sustained"] and round(d["sustained"]["seconds"],1), d["sustained"] and round(d["sustained"]["value_this_rank"],1),
    d["roofline"]["frac"], d["roofline"]["avg_launch_ms"]*1e3, d["roofline_latent_step"]["frac"], d["roofline_latent_step"]["fp32_equivalent"]["frac"],
    d.get("roofline_enh_front") and round(d["roofline_enh_front"]["frac"],3), d["exact_fp32_mode"]["value"], d["cpu_baseline"]["value"]))
PY
step "bench --gpus 4 --share-device"; timeout -k 10 300 python bench.py --gpus 4 --share-device --steps 10 --warmup 2 --no-exact --no-timer > $O/r5_bench_share4.json 2> $O/r5_bench_share4.err || { tail -n 20 $O/r5_bench_share4.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r5_bench_share4.json'));print('share4: %.1f scenes/s over %d ranks, %d scenes, group %s'%(d['value'],d['n_gpus'],d['total_scenes'],d['config']['process_group']))"
step "train leg --gpus 4 --share-device"; timeout -k 10 400 python bench.py --workload train --gpus 4 --share-device --steps 10 --warmup 3 > $O/r5_train_leg_share4.json 2> $O/r5_train_leg_share4.err || { tail -n 30 $O/r5_train_leg_share4.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r5_train_leg_share4.json'));print('train share4: %.1f scenes/s, %.1f ms/step, group %s, %d grad bytes/step'%(d['value'],d['ms_per_step'],d['config']['process_group'],d['config']['grad_bytes_allreduced_per_step']))"
step "train leg N=1"; timeout -k 10 400 python bench.py --workload train --steps 20 --warmup 3 > $O/r5_train_leg_n1_pre.json 2> $O/r5_train_leg_n1_pre.err || { tail -n 30 $O/r5_train_leg_n1_pre.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r5_train_leg_n1_pre.json'));print('train n1: %.1f scenes/s, %.1f ms/step, host enqueue %.1f ms'%(d['value'],d['ms_per_step'],d['host_enqueue_ms_per_step']))"
for wl in metric shipped; do
  step "bf16 $wl"; timeout -k 10 300 python bench.py --workload $wl --mode arith=2 --no-cpu-baseline --no-exact --sustain 2 > $O/r5_bench_bf16_$wl.json 2> $O/r5_bench_bf16_$wl.err || { tail -n 20 $O/r5_bench_bf16_$wl.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/r5_bench_bf16_$wl.json'));r=d.get('roofline') or {};print('bf16 $wl: %.1f scenes/s, latency %.3f ms, family frac %s'%(d['value'],d['latency_ms_one_scene'],r.get('frac')))"
done
for wl in shipped v2xreal; do
  step "bench $wl"; timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --sustain 2 > $O/r5_bench_${wl}_pre.json 2> $O/r5_bench_${wl}_pre.err || { tail -n 20 $O/r5_bench_${wl}_pre.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/r5_bench_${wl}_pre.json'));print('$wl: %.1f scenes/s, latency one scene %.3f ms'%(d['value'],d['latency_ms_one_scene']));print(d['kernel_time_shares'])"
done
echo done
