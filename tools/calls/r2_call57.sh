#!/bin/bash
# end of round: full GPU suite, smoke, whole-model / fusion / backbone / SECOND / training benches, default bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r2c57_pytest.log 2>&1
rc=$?
tail -n 4 gpurun_out/r2c57_pytest.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2
for f in att v2xvit where2comm; do timeout -k 10 300 python tools/shell_bench.py --fusion $f 2>&1 | grep -v amdgpu.ids | tail -2; done | tee gpurun_out/r2c57_shell.log
timeout -k 10 300 python tools/shell_bench.py --agents 5 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a gpurun_out/r2c57_shell.log
timeout -k 10 300 python tools/v2xvit_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c57_v2xvit.log
timeout -k 10 300 python tools/train_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c57_train.log
timeout -k 10 300 python tools/second_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c57_second.log
timeout -k 10 300 python tools/backbone_bench.py --n 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c57_backbone.log
timeout -k 10 300 python tools/backbone_bench.py --n 8 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2c57_backbone.log
timeout -k 10 400 python bench.py > gpurun_out/r2c57_bench.json 2> gpurun_out/r2c57_err.log || { tail gpurun_out/r2c57_err.log; exit 1; }
cut -c1-100 gpurun_out/r2c57_bench.json
