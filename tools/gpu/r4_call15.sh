#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_backward.py -m gpu -x -q > $O/r4c15_tests.log 2>&1 || { tail -n 40 $O/r4c15_tests.log; exit 1; }
tail -n 2 $O/r4c15_tests.log
timeout -k 10 400 python bench.py --workload train --steps 20 --warmup 3 > $O/r4c15_train1.json 2> $O/r4c15_train1.err || { tail -n 30 $O/r4c15_train1.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r4c15_train1.json'));print('train leg: %.1f scenes/s, %.1f ms/step'%(d['value'],d['ms_per_step']))"
rm -rf $O/r4c15_prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4c15_prof -o k -- python3 bench.py --workload train --steps 5 --warmup 2 > $O/r4c15_prof.log 2>&1 || { tail -n 20 $O/r4c15_prof.log; exit 1; }
find $O/r4c15_prof -name "*kernel_stats.csv" -exec cp {} $O/r4c15_leg_kernel_stats.csv \;
rm -rf $O/r4c15_prof
grep -E "dcn" $O/r4c15_leg_kernel_stats.csv | cut -c1-160
