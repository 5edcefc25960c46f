#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
echo "== tests"; timeout -k 10 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_backward.py tests/test_backbone.py -m gpu -x -q > $O/r4c9_tests.log 2>&1 || { tail -n 40 $O/r4c9_tests.log; exit 1; }
tail -n 3 $O/r4c9_tests.log
echo "== train leg"; timeout -k 10 400 python bench.py --workload train --steps 10 --warmup 3 > $O/r4c9_train1.json 2> $O/r4c9_train1.err || { tail -n 30 $O/r4c9_train1.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r4c9_train1.json'));print('train leg: %.1f scenes/s, %.1f ms/step'%(d['value'],d['ms_per_step']))"
echo "== profile train leg"; rm -rf $O/r4c9_prof_shell
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4c9_prof_shell -o k -- python3 bench.py --workload train --steps 5 --warmup 2 > $O/r4c9_prof_shell.log 2>&1 || { tail -n 20 $O/r4c9_prof_shell.log; exit 1; }
find $O/r4c9_prof_shell -name "*kernel_stats.csv" -exec cp {} $O/r4c9_shell_kernel_stats.csv \;
T=$(find $O/r4c9_prof_shell -name "*kernel_trace.csv" | head -1)
python tools/trace_by_grid.py $T > $O/r4c9_shell_by_grid.csv
rm -rf $O/r4c9_prof_shell
head -n 22 $O/r4c9_shell_kernel_stats.csv | cut -c1-150
echo done
