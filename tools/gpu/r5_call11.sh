#!/bin/bash
# Round 5, call 11: the training leg under GENCOMM_MODE_BWD_STREAMS = 1 (default) / 0 (one stream) / 2 (side stream on every call), twice each,
# and on the exact-fp32 kernels (GENCOMM_MODE_ARITH = 1) -- `--mode` reaches the leg only since the end of round 5
set -o pipefail
O=gpurun_out; mkdir -p $O
for m in bwd_streams=1 bwd_streams=0 bwd_streams=2 bwd_streams=1 bwd_streams=0 bwd_streams=2 arith=1; do
  timeout -k 10 280 python bench.py --workload train --steps 20 --warmup 3 --no-cpu-baseline --mode $m > $O/r5_train_mode.json 2> $O/r5_train_mode.err || { tail -n 20 $O/r5_train_mode.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/r5_train_mode.json')); print('$m: %.2f ms/step, host %.2f ms, modes %s'%(d['ms_per_step'], d['host_enqueue_ms_per_step'], d['config']['modes']))"
done 2>&1 | tee $O/r5_train_stream_modes.txt
