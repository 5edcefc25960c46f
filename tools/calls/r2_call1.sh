#!/bin/bash
# round 2, GPU call 1: full GPU suite with the new noise generator + XCD mapping, bench A/B, kernel trace
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > gpurun_out/r2c1_pytest.log 2>&1
rc=$?
tail -n 40 gpurun_out/r2c1_pytest.log
echo "pytest rc=$rc"
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > gpurun_out/r2c1_bench_xcd1.json 2> gpurun_out/r2c1_bench_xcd1.err || exit 1
tail -c 600 gpurun_out/r2c1_bench_xcd1.json | head -c 0
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c1_bench_xcd1.json'))
print('xcd=1', d['value'], d['roofline']['avg_launch_ms'] if d.get('roofline') else None, d.get('roofline_conv16',{}).get('avg_launch_ms'), d.get('exact_fp32_mode',{}).get('value'), d.get('cpu_baseline',{}).get('value'))
PY
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --mode xcd=0 > gpurun_out/r2c1_bench_xcd0.json 2> gpurun_out/r2c1_bench_xcd0.err || exit 1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c1_bench_xcd0.json'))
print('xcd=0', d['value'], d['roofline']['avg_launch_ms'] if d.get('roofline') else None, d.get('roofline_conv16',{}).get('avg_launch_ms'), d.get('exact_fp32_mode',{}).get('value'))
PY
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r2c1_prof -o r2c1 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --streams 1 > gpurun_out/r2c1_prof.log 2>&1 || exit 1
ls gpurun_out/r2c1_prof | head
