#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
echo "== rocprof metric 1 stream"; rm -rf $O/r4_prof_metric
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_prof_metric -o k -- python3 bench.py --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-exact --no-timer > $O/r4_prof_metric.log 2>&1 || { tail -n 20 $O/r4_prof_metric.log; exit 1; }
find $O/r4_prof_metric -name "*kernel_stats.csv" -exec cp {} $O/r4_bench_metric_1stream_kernel_stats.csv \;
python tools/trace_by_grid.py $(find $O/r4_prof_metric -name "*kernel_trace.csv" | head -1) > $O/r4_bench_metric_1stream_kernel_stats_by_grid.csv
rm -rf $O/r4_prof_metric
head -n 8 $O/r4_bench_metric_1stream_kernel_stats.csv | cut -c1-140
echo "== pmc traffic"; bash tools/pmc_pass.sh metric > $O/r4_pmc_pass.log 2>&1 || { tail -n 20 $O/r4_pmc_pass.log; exit 1; }
echo "== pmc sq"; bash tools/pmc_sq_pass.sh metric > $O/r4_pmc_sq_pass.log 2>&1 || { tail -n 20 $O/r4_pmc_sq_pass.log; exit 1; }
echo "== pmc insts"; bash tools/pmc_insts_pass.sh metric > $O/r4_pmc_insts_pass.log 2>&1 || { tail -n 20 $O/r4_pmc_insts_pass.log; exit 1; }
rm -rf $O/r4_pmc_FETCH_SIZE $O/r4_pmc_WRITE_SIZE $O/r4_pmc_SQ $O/r4_pmc_INSTS
cp $O/r4_pmc_traffic.json profiles/r4_pmc_traffic.json
timeout -k 10 400 python bench.py --no-cpu-baseline --no-exact > $O/r4_bench_check.json 2> $O/r4_bench_check.err || { tail -n 20 $O/r4_bench_check.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r4_bench_check.json'));print('check: %.1f scenes/s, frac %.3f, traffic %s (%s) alg %s'%(d['value'],d['roofline']['frac'],d['roofline']['traffic'],d['roofline']['traffic_source'],d['roofline']['algorithmic_bytes_per_launch']))"
