#!/bin/bash
# End-of-round reference measurements, part B (round 5): the training leg (RCCL world 1, flat bucket and DistributedDataParallel; gloo share-device
# world 4), the hot-path training step, rocprofv3 kernel statistics of both (the leg also per grid), the general convolution per shape.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
step() { echo "== $1"; }
step "train leg N=1 (RCCL, world 1)"; timeout -k 10 400 python bench.py --workload train --steps 20 --warmup 3 > $O/r5_train_leg_n1.json 2> $O/r5_train_leg_n1.err || { tail -n 30 $O/r5_train_leg_n1.err; exit 1; }
step "train leg N=1 under torch DistributedDataParallel"; timeout -k 10 400 python bench.py --workload train --grad-sync ddp --steps 20 --warmup 3 --no-cpu-baseline > $O/r5_train_leg_n1_ddp.json 2> $O/r5_train_leg_n1_ddp.err || { tail -n 30 $O/r5_train_leg_n1_ddp.err; exit 1; }
step "train leg N=4 share-device (gloo)"; timeout -k 10 400 python bench.py --workload train --gpus 4 --share-device --steps 10 --warmup 3 --no-cpu-baseline > $O/r5_train_leg_share4.json 2> $O/r5_train_leg_share4.err || { tail -n 30 $O/r5_train_leg_share4.err; exit 1; }
step "train leg N=1, exact fp32 (GENCOMM_MODE_ARITH = 1)"; timeout -k 10 400 python bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline --mode arith=1 > $O/r5_train_leg_n1_exact.json 2> $O/r5_train_leg_n1_exact.err || { tail -n 30 $O/r5_train_leg_n1_exact.err; exit 1; }
python - <<'PY'
import json
for f in ("n1","n1_ddp","share4","n1_exact"):
    d=json.load(open(f"gpurun_out/r5_train_leg_{f}.json"))
    r=d.get("roofline") or {}
    print("train leg %s: %.1f scenes/s, %.2f ms/step (host %.2f), group %s rccl_ranks %d, %d grad bytes/step; family %s launches, %.1f TFLOP/s, frac %s"%(f,d["value"],d["ms_per_step"],d["host_enqueue_ms_per_step"],d["config"]["process_group"],d["config"]["rccl_ranks"],d["config"]["grad_bytes_allreduced_per_step"], r.get("launches"), r.get("achieved",0.0), r.get("frac")))
d=json.load(open("gpurun_out/r5_train_leg_n1.json")); print("cpu_baseline:", d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"])
PY
step "train_bench"; : > $O/r5_train_step.txt
for b in 1 2 4; do timeout -k 10 300 python tools/train_bench.py --batch $b 2>&1 | grep "train step" >> $O/r5_train_step.txt || exit 1; done
cat $O/r5_train_step.txt
step "general convolution per shape"; timeout -k 10 300 python tools/conv_h3_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/r5_conv_h3_bench.txt
step "rocprof train step (large, batch 1)"; rm -rf $O/r5_prof_train
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_prof_train -o k -- python3 tools/train_bench.py --only large > $O/r5_prof_train.log 2>&1 || { tail -n 20 $O/r5_prof_train.log; exit 1; }
find $O/r5_prof_train -name "*kernel_stats.csv" -exec cp {} $O/r5_train_step_kernel_stats.csv \;
rm -rf $O/r5_prof_train
step "rocprof train leg"; rm -rf $O/r5_prof_leg
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_prof_leg -o k -- python3 bench.py --workload train --steps 5 --warmup 2 --no-cpu-baseline --no-timer > $O/r5_prof_leg.log 2>&1 || { tail -n 20 $O/r5_prof_leg.log; exit 1; }
find $O/r5_prof_leg -name "*kernel_stats.csv" -exec cp {} $O/r5_train_leg_kernel_stats.csv \;
python tools/trace_by_grid.py $(find $O/r5_prof_leg -name "*kernel_trace.csv" | head -1) > $O/r5_train_leg_kernel_stats_by_grid.csv
rm -rf $O/r5_prof_leg
head -n 16 $O/r5_train_leg_kernel_stats_by_grid.csv | cut -c1-170
echo done
