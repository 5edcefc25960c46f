#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c61_attn.log
