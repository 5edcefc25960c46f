#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
rm -rf $O/r4c7_prof_train
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/r4c7_prof_train -o k -- python3 tools/train_bench.py --only large > $O/r4c7_prof_train.log 2>&1 || { tail -n 20 $O/r4c7_prof_train.log; exit 1; }
T=$(find $O/r4c7_prof_train -name "*kernel_trace.csv" | head -1)
python tools/trace_by_grid.py $T > $O/r4c7_train_by_grid.csv
python tools/trace_by_grid.py $T wgrad gn_silu dwconv lincomb ln_nchw > $O/r4c7_train_by_grid_sel.csv
rm -rf $O/r4c7_prof_train
head -n 40 $O/r4c7_train_by_grid_sel.csv | cut -c1-190
