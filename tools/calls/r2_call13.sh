#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_second.py tests/test_shell.py -m gpu -q -x > gpurun_out/r2c13_pytest.log 2>&1
rc=$?
tail -n 30 gpurun_out/r2c13_pytest.log | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/second_bench.py > gpurun_out/r2c13_second.log 2>&1 || { tail -n 20 gpurun_out/r2c13_second.log; exit 1; }
cat gpurun_out/r2c13_second.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r2c13_prof -o sec -- python3 tools/second_bench.py > gpurun_out/r2c13_prof.log 2>&1 || { tail gpurun_out/r2c13_prof.log; exit 1; }
python tools/rocpd_stats.py gpurun_out/r2c13_prof/sec_results.db > gpurun_out/r2c13_second_kernel_stats.csv
rm -rf gpurun_out/r2c13_prof
head -n 16 gpurun_out/r2c13_second_kernel_stats.csv | cut -c1-170
