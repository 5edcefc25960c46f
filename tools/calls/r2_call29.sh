#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_properties.py -m gpu -q -x > gpurun_out/r2c29_pytest.log 2>&1
rc=$?
tail -n 6 gpurun_out/r2c29_pytest.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --no-exact > gpurun_out/r2c29_bench.json 2> gpurun_out/r2c29_err.log || { tail gpurun_out/r2c29_err.log; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c29_bench.json'))
print('value', round(d['value'],2), {k[:28]: v for k, v in d['kernel_time_shares'].items() if 'enh' in k or 'gemm' in k})
PY
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r2c29_prof -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exact --no-timer --streams 1 > gpurun_out/r2c29_prof.log 2>&1 || exit 1
python tools/rocpd_stats.py gpurun_out/r2c29_prof/p_results.db > gpurun_out/r2c29_stats.csv; rm -rf gpurun_out/r2c29_prof
grep "enh_\|gemm_\|warp_att" gpurun_out/r2c29_stats.csv | cut -c1-140
