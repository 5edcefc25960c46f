#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for i in 1 2; do timeout -k 10 300 python tools/v2xvit_bench.py 2>&1 | grep -v amdgpu.ids | grep "v2xvit\|where2comm: 2"; done
