#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
echo "== tests"; timeout -k 10 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_backward.py tests/test_v2xvit.py tests/test_where2comm.py -m gpu -x -q > $O/r4c12_tests.log 2>&1 || { tail -n 40 $O/r4c12_tests.log; exit 1; }
tail -n 2 $O/r4c12_tests.log
echo "== train_bench"; : > $O/r4c12_train_step.txt
for b in 1 4; do timeout -k 10 300 python tools/train_bench.py --batch $b 2>&1 | grep "train step" >> $O/r4c12_train_step.txt || exit 1; done
cat $O/r4c12_train_step.txt
rm -rf $O/r4c12_prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4c12_prof -o k -- python3 tools/train_bench.py --only large > $O/r4c12_prof.log 2>&1 || { tail -n 20 $O/r4c12_prof.log; exit 1; }
find $O/r4c12_prof -name "*kernel_stats.csv" -exec cp {} $O/r4c12_train_kernel_stats.csv \;
rm -rf $O/r4c12_prof
grep -E "dwconv|ln_nchw" $O/r4c12_train_kernel_stats.csv | cut -c1-150
