#!/bin/bash
# round 2, the round's reference measurements (run again after every batch of changes) -- full GPU suite, default bench line, 1-stream kernel trace,
# PMC traffic passes, bf16 / shipped / cfg2 lines, train step, V2X-ViT latency
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r2c11_pytest.log 2>&1
rc=$?
tail -n 12 gpurun_out/r2c11_pytest.log | cut -c1-240
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/r2c11_bench.json 2> gpurun_out/r2c11_bench.err || { tail -n 20 gpurun_out/r2c11_bench.err; exit 1; }
cut -c1-600 gpurun_out/r2c11_bench.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r2c11_prof -o r2c11 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact --no-timer --streams 1 > gpurun_out/r2c11_prof.log 2>&1 || { tail -n 20 gpurun_out/r2c11_prof.log; exit 1; }
python tools/rocpd_stats.py gpurun_out/r2c11_prof/r2c11_results.db > gpurun_out/r2c11_kernel_stats.csv && python tools/rocpd_stats.py gpurun_out/r2c11_prof/r2c11_results.db --by-grid > gpurun_out/r2c11_kernel_stats_by_grid.csv
rm -rf gpurun_out/r2c11_prof
head -n 8 gpurun_out/r2c11_kernel_stats.csv | cut -c1-160
bash tools/pmc_pass.sh metric > gpurun_out/r2c11_pmc.log 2>&1 || { tail -n 20 gpurun_out/r2c11_pmc.log; exit 1; }
for wl in shipped cfg2 v2xreal; do
  timeout -k 10 300 python bench.py --workload $wl --no-exact > gpurun_out/r2c11_bench_$wl.json 2> gpurun_out/r2c11_bench_$wl.err || exit 1
  cut -c1-200 gpurun_out/r2c11_bench_$wl.json
done
timeout -k 10 300 python bench.py --mode arith=2 --no-cpu-baseline --no-exact > gpurun_out/r2c11_bench_bf16.json 2> gpurun_out/r2c11_bench_bf16.err || exit 1
cut -c1-200 gpurun_out/r2c11_bench_bf16.json
timeout -k 10 300 python tools/train_bench.py > gpurun_out/r2c11_train.log 2>&1 || exit 1
cat gpurun_out/r2c11_train.log
timeout -k 10 300 python tools/v2xvit_bench.py > gpurun_out/r2c11_v2xvit.log 2>&1 || { tail gpurun_out/r2c11_v2xvit.log; exit 1; }
cat gpurun_out/r2c11_v2xvit.log
timeout -k 10 300 python tools/second_bench.py > gpurun_out/r2c11_second.log 2>&1 || { tail gpurun_out/r2c11_second.log; exit 1; }
cat gpurun_out/r2c11_second.log
timeout -k 10 300 python bench.py --workload shipped --batch 1 --streams 1 --steps 50 --warmup 5 --no-cpu-baseline --no-exact --no-timer > gpurun_out/r2c11_bench_shipped_1x1.json 2> gpurun_out/r2c11_bench_shipped_1x1.err || exit 1
cut -c1-200 gpurun_out/r2c11_bench_shipped_1x1.json
