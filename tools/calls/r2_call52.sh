#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for f in att v2xvit where2comm; do timeout -k 10 300 python tools/shell_bench.py --fusion $f 2>&1 | grep -v amdgpu.ids | tail -3; done | tee gpurun_out/r2c52_shell.log
timeout -k 10 300 python tools/shell_bench.py --agents 5 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a gpurun_out/r2c52_shell.log
