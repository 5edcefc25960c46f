#!/bin/bash
# round-4 call 1: new tests, self-launching bench (share-device rehearsal), training leg (RCCL at world 1, gloo at 2), training-step profile at HEAD
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== tests"; timeout -k 10 600 python -m pytest tests/test_loss.py tests/test_gpu_backward.py -m gpu -x -q -k "loss or validates or two_rank or stage1_training" > $O/r4c1_tests.log 2>&1 || { tail -n 30 $O/r4c1_tests.log; exit 1; }
tail -n 3 $O/r4c1_tests.log
echo "== bench default"; timeout -k 10 400 python bench.py > $O/r4c1_bench.json 2> $O/r4c1_bench.err || { tail -n 20 $O/r4c1_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c1_bench.json"))
print("bench: %.1f scenes/s ms/step %.2f latency1 %.2f sustained %s frac %.3f traffic %s latent %.3f ms exact %.1f cpu %.4f" % (d["value"], d["ms_per_step"], d["latency_ms_one_scene"], d["sustained"] and round(d["sustained"]["value_this_rank"],1), d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline_latent_step"]["avg_launch_ms"], d["exact_fp32_mode"]["value"], d["cpu_baseline"]["value"]))
print(json.dumps(d["kernel_time_shares"]))
PY
echo "== bench --gpus 2 --share-device"; timeout -k 10 300 python bench.py --gpus 2 --share-device --steps 10 --warmup 2 --no-cpu-baseline --no-exact --no-timer --streams 2 > $O/r4c1_share2.json 2> $O/r4c1_share2.err || { tail -n 20 $O/r4c1_share2.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r4c1_share2.json'));print('share2:',d['value'],d['n_gpus'],d['total_scenes'],d['config']['process_group'],d['config']['rccl_ranks'])"
echo "== train N=1 (nccl world 1)"; timeout -k 10 400 python bench.py --workload train --steps 10 --warmup 3 > $O/r4c1_train1.json 2> $O/r4c1_train1.err || { tail -n 30 $O/r4c1_train1.err; exit 1; }
cat $O/r4c1_train1.json
echo "== train N=2 share-device (gloo)"; timeout -k 10 400 python bench.py --workload train --gpus 2 --share-device --steps 6 --warmup 2 > $O/r4c1_train2.json 2> $O/r4c1_train2.err || { tail -n 30 $O/r4c1_train2.err; exit 1; }
cat $O/r4c1_train2.json
echo "== train_bench"; timeout -k 10 300 python tools/train_bench.py > $O/r4c1_train_bench.txt 2>&1 || { tail -n 20 $O/r4c1_train_bench.txt; exit 1; }
timeout -k 10 200 python tools/train_bench.py --only large --batch 4 >> $O/r4c1_train_bench.txt 2>&1
cat $O/r4c1_train_bench.txt
echo "== profile train step (large)"; rm -rf $O/r4c1_prof_train
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4c1_prof_train -o k -- python3 tools/train_bench.py --only large > $O/r4c1_prof_train.log 2>&1 || { tail -n 20 $O/r4c1_prof_train.log; exit 1; }
find $O/r4c1_prof_train -name "*kernel_stats.csv" -exec cp {} $O/r4c1_train_kernel_stats.csv \;
find $O/r4c1_prof_train -name "*kernel_trace.csv" -delete
head -n 40 $O/r4c1_train_kernel_stats.csv
echo "== profile train leg (stage-1 shell)"; rm -rf $O/r4c1_prof_shell
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4c1_prof_shell -o k -- python3 bench.py --workload train --steps 5 --warmup 2 > $O/r4c1_prof_shell.log 2>&1 || { tail -n 20 $O/r4c1_prof_shell.log; exit 1; }
find $O/r4c1_prof_shell -name "*kernel_stats.csv" -exec cp {} $O/r4c1_shell_kernel_stats.csv \;
find $O/r4c1_prof_shell -name "*kernel_trace.csv" -delete
head -n 30 $O/r4c1_shell_kernel_stats.csv
echo done
