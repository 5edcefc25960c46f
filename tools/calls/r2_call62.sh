#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "attn" > gpurun_out/r2c62_pytest.log 2>&1
rc=$?
tail -n 15 gpurun_out/r2c62_pytest.log | cut -c1-250
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c62_attn.log
