#!/bin/bash
# new LayerNorm + split-MFMA 1x1 convolution: everything outside the UNet tests, then the fusion latency bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_conv8.py --deselect tests/test_gpu_properties.py --deselect tests/test_gpu_configs.py > gpurun_out/r2c41_pytest.log 2>&1
rc=$?
tail -n 25 gpurun_out/r2c41_pytest.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/v2xvit_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c41_v2xvit.log
