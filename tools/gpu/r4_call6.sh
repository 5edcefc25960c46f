#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== tests"; timeout -k 10 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_backward.py tests/test_ap_chain.py -m gpu -x -q > $O/r4c6_tests.log 2>&1 || { tail -n 40 $O/r4c6_tests.log; exit 1; }
tail -n 3 $O/r4c6_tests.log
echo "== train_bench"; timeout -k 10 300 python tools/train_bench.py > $O/r4c6_train_bench.txt 2>&1 || { tail -n 20 $O/r4c6_train_bench.txt; exit 1; }
timeout -k 10 200 python tools/train_bench.py --only large --batch 4 >> $O/r4c6_train_bench.txt 2>&1
grep "train step" $O/r4c6_train_bench.txt
echo "== profile train step (large)"; rm -rf $O/r4c6_prof_train
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4c6_prof_train -o k -- python3 tools/train_bench.py --only large > $O/r4c6_prof_train.log 2>&1 || { tail -n 20 $O/r4c6_prof_train.log; exit 1; }
find $O/r4c6_prof_train -name "*kernel_stats.csv" -exec cp {} $O/r4c6_train_kernel_stats.csv \;
find $O/r4c6_prof_train -name "*kernel_trace.csv" -delete
head -n 12 $O/r4c6_train_kernel_stats.csv | cut -c1-160
echo "== train leg"; timeout -k 10 400 python bench.py --workload train --steps 10 --warmup 3 > $O/r4c6_train1.json 2> $O/r4c6_train1.err || { tail -n 30 $O/r4c6_train1.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r4c6_train1.json'));print('train leg: %.1f scenes/s, %.1f ms/step'%(d['value'],d['ms_per_step']))"
echo done
