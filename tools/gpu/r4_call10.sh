#!/bin/bash
set -o pipefail
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_enhancer_fuse.py tests/test_gpu_backward.py tests/test_gpu_parity.py -m gpu -x -q -s > $O/r4c10_tests.log 2>&1 || { tail -n 40 $O/r4c10_tests.log; exit 1; }
grep "large weights" $O/r4c10_tests.log; tail -n 2 $O/r4c10_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-exact > $O/r4c10_bench.json 2> $O/r4c10_bench.err || { tail -n 20 $O/r4c10_bench.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r4c10_bench.json'));print('bench: %.1f scenes/s; shares %s'%(d['value'], {k[:24]:v for k,v in list(d['kernel_time_shares'].items())[:8]}))"
