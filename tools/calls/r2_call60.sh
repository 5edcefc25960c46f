#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_backbone.py tests/test_gpu_backward.py tests/test_late.py tests/test_second.py -m gpu -q -x > gpurun_out/r2c60_pytest.log 2>&1
rc=$?
tail -n 4 gpurun_out/r2c60_pytest.log | cut -c1-250
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/train_bench.py 2>&1 | grep -v amdgpu.ids
