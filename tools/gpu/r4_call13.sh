#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
echo "== tests"; timeout -k 10 1100 python -m pytest tests/test_backbone.py tests/test_shell.py tests/test_gpu_train_ops.py tests/test_gpu_backward.py tests/test_v2xvit.py tests/test_where2comm.py tests/test_second.py tests/test_late.py tests/test_ap_chain.py tests/test_msgext_oracle.py -m gpu -x -q > $O/r4c13_tests.log 2>&1 || { tail -n 40 $O/r4c13_tests.log; exit 1; }
tail -n 2 $O/r4c13_tests.log
echo "== backbone"; timeout -k 10 200 python tools/backbone_bench.py --n 2 2>&1 | grep -v amdgpu | tail -4
echo "== shell"; timeout -k 10 200 python tools/shell_bench.py 2>&1 | grep -v amdgpu | tail -3
echo "== train leg"; timeout -k 10 400 python bench.py --workload train --steps 20 --warmup 3 > $O/r4c13_train1.json 2> $O/r4c13_train1.err || { tail -n 30 $O/r4c13_train1.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r4c13_train1.json'));print('train leg: %.1f scenes/s, %.1f ms/step'%(d['value'],d['ms_per_step']))"
timeout -k 10 300 python tools/train_bench.py --only large 2>&1 | grep "train step"
