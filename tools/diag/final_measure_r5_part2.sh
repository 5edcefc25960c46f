#!/bin/bash
# End-of-round reference measurements on ONE box (round 5): bench lines of the four workloads, the self-launched 2-rank rehearsal, the training
# leg (RCCL world 1; gloo share-device world 2), rocprofv3 kernel statistics of the metric workload (one stream), of the large training step
# and of the training leg, the PMC passes (traffic -> stamped with the library's source hash; SQ; instruction mix), the parity margins
# of the metric-size chain.  Outputs under gpurun_out/r5_*; the summaries are then copied to profiles/.
set -o pipefail
export TMPDIR=/tmp
export ROUND=r5
O=gpurun_out; mkdir -p $O
step() { echo "== $1"; }
step "train leg N=1 (RCCL, world 1)"; timeout -k 10 400 python bench.py --workload train --steps 20 --warmup 3 > $O/r5_train_leg_n1.json 2> $O/r5_train_leg_n1.err || { tail -n 30 $O/r5_train_leg_n1.err; exit 1; }
step "train leg N=1 under torch DistributedDataParallel"; timeout -k 10 400 python bench.py --workload train --grad-sync ddp --steps 20 --warmup 3 > $O/r5_train_leg_n1_ddp.json 2> $O/r5_train_leg_n1_ddp.err || { tail -n 30 $O/r5_train_leg_n1_ddp.err; exit 1; }
step "train leg N=4 share-device (gloo)"; timeout -k 10 400 python bench.py --workload train --gpus 4 --share-device --steps 10 --warmup 3 > $O/r5_train_leg_share4.json 2> $O/r5_train_leg_share4.err || { tail -n 30 $O/r5_train_leg_share4.err; exit 1; }
python - <<'PY'
import json
for f in ("n1","n1_ddp","share4"):
    d=json.load(open(f"gpurun_out/r5_train_leg_{f}.json"))
    print("train leg %s: %.1f scenes/s, %.1f ms/step, group %s rccl_ranks %d, %d grad bytes/step"%(f,d["value"],d["ms_per_step"],d["config"]["process_group"],d["config"]["rccl_ranks"],d["config"]["grad_bytes_allreduced_per_step"]))
PY
for wl in metric shipped; do
  step "bf16 $wl"; timeout -k 10 300 python bench.py --workload $wl --mode arith=2 --no-cpu-baseline --no-exact --sustain 2 > $O/r5_bench_bf16_$wl.json 2> $O/r5_bench_bf16_$wl.err || { tail -n 20 $O/r5_bench_bf16_$wl.err; exit 1; }
done
step "train_bench"; : > $O/r5_train_step.txt
for b in 1 2 4; do timeout -k 10 300 python tools/train_bench.py --batch $b 2>&1 | grep "train step" >> $O/r5_train_step.txt || exit 1; done
cat $O/r5_train_step.txt
step "rocprof metric 1 stream"; rm -rf $O/r5_prof_metric
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_prof_metric -o k -- python3 bench.py --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-exact --no-timer > $O/r5_prof_metric.log 2>&1 || { tail -n 20 $O/r5_prof_metric.log; exit 1; }
find $O/r5_prof_metric -name "*kernel_stats.csv" -exec cp {} $O/r5_bench_metric_1stream_kernel_stats.csv \;
python tools/trace_by_grid.py $(find $O/r5_prof_metric -name "*kernel_trace.csv" | head -1) > $O/r5_bench_metric_1stream_kernel_stats_by_grid.csv
rm -rf $O/r5_prof_metric
step "rocprof train step (large, batch 1)"; rm -rf $O/r5_prof_train
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_prof_train -o k -- python3 tools/train_bench.py --only large > $O/r5_prof_train.log 2>&1 || { tail -n 20 $O/r5_prof_train.log; exit 1; }
find $O/r5_prof_train -name "*kernel_stats.csv" -exec cp {} $O/r5_train_step_kernel_stats.csv \;
rm -rf $O/r5_prof_train
step "rocprof train leg"; rm -rf $O/r5_prof_leg
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_prof_leg -o k -- python3 bench.py --workload train --steps 5 --warmup 2 > $O/r5_prof_leg.log 2>&1 || { tail -n 20 $O/r5_prof_leg.log; exit 1; }
find $O/r5_prof_leg -name "*kernel_stats.csv" -exec cp {} $O/r5_train_leg_kernel_stats.csv \;
rm -rf $O/r5_prof_leg
step "pmc traffic"; bash tools/pmc_pass.sh metric > $O/r5_pmc_pass.log 2>&1 || { tail -n 20 $O/r5_pmc_pass.log; exit 1; }
step "pmc sq"; bash tools/pmc_sq_pass.sh metric > $O/r5_pmc_sq_pass.log 2>&1 || { tail -n 20 $O/r5_pmc_sq_pass.log; exit 1; }
step "pmc insts"; bash tools/pmc_insts_pass.sh metric > $O/r5_pmc_insts_pass.log 2>&1 || { tail -n 20 $O/r5_pmc_insts_pass.log; exit 1; }
rm -rf $O/r5_pmc_FETCH_SIZE $O/r5_pmc_WRITE_SIZE $O/r5_pmc_SQ $O/r5_pmc_INSTS
step "parity margins"; timeout -k 10 600 python -m pytest tests/test_gpu_philox_replay.py tests/test_gpu_configs.py -m gpu -q -s -k "metric" > $O/r5_parity.log 2>&1 || { tail -n 20 $O/r5_parity.log; exit 1; }
grep -E "vs float64|vs float32 oracle" $O/r5_parity.log > $O/r5_parity_margins.txt; cat $O/r5_parity_margins.txt | cut -c1-260
step "bench default again (traffic and issue now stamped)"; cp $O/r5_pmc_traffic.json profiles/r5_pmc_traffic.json; cp $O/r5_pmc_sq.json profiles/r5_pmc_sq.json
timeout -k 10 400 python bench.py > $O/r5_bench_default.json 2> $O/r5_bench_default.err || { tail -n 20 $O/r5_bench_default.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r5_bench_default.json'));print('default (2nd): %.1f scenes/s, frac %.3f, traffic %s (%s)'%(d['value'],d['roofline']['frac'],d['roofline']['traffic'],d['roofline']['traffic_source']))"
echo done
