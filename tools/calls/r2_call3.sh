#!/bin/bash
# round 2, GPU call 3: full suite after the range guard / downsample kernel / test policy; bench; kernel trace (csv)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > gpurun_out/r2c3_pytest.log 2>&1
rc=$?
grep -n "worst err/tol\|vs float64\|large inputs" gpurun_out/r2c3_pytest.log | cut -c1-260 | tail -n 60
tail -n 15 gpurun_out/r2c3_pytest.log
echo "pytest rc=$rc"
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > gpurun_out/r2c3_bench.json 2> gpurun_out/r2c3_bench.err || exit 1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c3_bench.json'))
print('bench', d['value'], d['roofline']['avg_launch_ms'] if d.get('roofline') else None, d.get('roofline_conv16',{}).get('avg_launch_ms'), d.get('exact_fp32_mode',{}).get('value'), d.get('cpu_baseline',{}).get('value'))
PY
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r2c3_prof -o r2c3 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --streams 1 > gpurun_out/r2c3_prof.log 2>&1 || exit 1
python tools/rocpd_stats.py gpurun_out/r2c3_prof/r2c3_results.db --by-grid > gpurun_out/r2c3_kernel_stats_by_grid.csv
head -n 12 gpurun_out/r2c3_kernel_stats_by_grid.csv | cut -c1-150
