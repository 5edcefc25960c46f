#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ap_chain.py -m gpu -x -q -s > $O/r4c5_apchain.log 2>&1 || { tail -n 40 $O/r4c5_apchain.log; exit 1; }
grep "AP chain" $O/r4c5_apchain.log; tail -n 2 $O/r4c5_apchain.log
echo "== train N=1 (bucket view)"; timeout -k 10 400 python bench.py --workload train --steps 10 --warmup 3 > $O/r4c5_train1.json 2> $O/r4c5_train1.err || { tail -n 30 $O/r4c5_train1.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r4c5_train1.json'));print('train: %.1f scenes/s, %.1f ms/step'%(d['value'],d['ms_per_step']))"
echo done
