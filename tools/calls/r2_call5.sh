#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_iou3d_voxel.py tests/test_backbone.py -m gpu -q -x > gpurun_out/r2c5_pytest.log 2>&1
rc=$?
tail -n 40 gpurun_out/r2c5_pytest.log
echo "pytest rc=$rc"
