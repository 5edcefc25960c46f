#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/train_bench.py > gpurun_out/r2c17_train.log 2>&1 || { tail gpurun_out/r2c17_train.log; exit 1; }
cat gpurun_out/r2c17_train.log
timeout -k 10 300 python tools/train_bench.py --no-optimizer > gpurun_out/r2c17_train_noopt.log 2>&1 || { tail gpurun_out/r2c17_train_noopt.log; exit 1; }
cat gpurun_out/r2c17_train_noopt.log
