#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py "tests/test_gpu_parity.py::test_enhancer_and_fusion_gradients_flow" tests/test_gpu_bf16.py -m gpu -q -s > gpurun_out/r2c8_pytest.log 2>&1
rc=$?
grep -n "worst relative\|passed\|failed" gpurun_out/r2c8_pytest.log | cut -c1-200
tail -n 12 gpurun_out/r2c8_pytest.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python tools/train_bench.py 2>&1 | grep "train step"
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-exact --mode arith=2 > gpurun_out/r2c8_bench_bf16.json 2> gpurun_out/r2c8_bench_bf16.err || { tail -n 20 gpurun_out/r2c8_bench_bf16.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c8_bench_bf16.json')); r=d['roofline']
print('bf16 value', d['value'])
for v in r['variants']: print('   %-32s %.1f us' % (v['variant'], 1e3*v['avg_launch_ms']))
PY
