#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_conv8.py --deselect tests/test_gpu_properties.py --deselect tests/test_gpu_configs.py > gpurun_out/r2c44_pytest.log 2>&1
rc=$?
tail -n 25 gpurun_out/r2c44_pytest.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/backbone_bench.py --n 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c44_backbone.log
timeout -k 10 300 python tools/backbone_bench.py --n 8 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2c44_backbone.log
