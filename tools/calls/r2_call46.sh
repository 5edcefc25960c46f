#!/bin/bash
# end-of-round regression: full GPU suite, smoke(), then the secondary benches touched by the convolution / attention kernels
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r2c46_pytest.log 2>&1
rc=$?
tail -n 5 gpurun_out/r2c46_pytest.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3
timeout -k 10 300 python tools/v2xvit_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c46_v2xvit.log
timeout -k 10 300 python tools/train_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c46_train.log
timeout -k 10 300 python tools/second_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c46_second.log
timeout -k 10 300 python tools/backbone_bench.py --n 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c46_backbone.log
timeout -k 10 300 python tools/backbone_bench.py --n 8 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2c46_backbone.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-exact > gpurun_out/r2c46_bench.json 2> gpurun_out/r2c46_err.log || { tail gpurun_out/r2c46_err.log; exit 1; }
cut -c1-100 gpurun_out/r2c46_bench.json
