#!/bin/bash
# End-of-round reference measurements, part C (round 5): rocprofv3 kernel statistics of the metric workload (one stream, also per grid), the PMC
# passes (traffic -> stamped with the library's source hash; SQ; instruction mix), the parity margins of the metric-size chain, and the
# default bench line once more with the fresh stamps in place.  Copies the stamped JSONs into profiles/ ON THE BOX for that last run; copy
# them into the repository's profiles/ afterwards (gpurun_out/r5_pmc_*.json).
set -o pipefail
export TMPDIR=/tmp
export ROUND=r5
O=gpurun_out; mkdir -p $O
step() { echo "== $1"; }
step "rocprof metric 1 stream"; rm -rf $O/r5_prof_metric
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r5_prof_metric -o k -- python3 bench.py --steps 5 --warmup 2 --streams 1 --no-cpu-baseline --no-exact --no-timer > $O/r5_prof_metric.log 2>&1 || { tail -n 20 $O/r5_prof_metric.log; exit 1; }
find $O/r5_prof_metric -name "*kernel_stats.csv" -exec cp {} $O/r5_bench_metric_1stream_kernel_stats.csv \;
python tools/trace_by_grid.py $(find $O/r5_prof_metric -name "*kernel_trace.csv" | head -1) > $O/r5_bench_metric_1stream_kernel_stats_by_grid.csv
rm -rf $O/r5_prof_metric
head -n 14 $O/r5_bench_metric_1stream_kernel_stats_by_grid.csv | cut -c1-150
step "pmc traffic"; bash tools/pmc_pass.sh metric > $O/r5_pmc_pass.log 2>&1 || { tail -n 20 $O/r5_pmc_pass.log; exit 1; }
step "pmc sq"; bash tools/pmc_sq_pass.sh metric > $O/r5_pmc_sq_pass.log 2>&1 || { tail -n 20 $O/r5_pmc_sq_pass.log; exit 1; }
step "pmc insts"; bash tools/pmc_insts_pass.sh metric > $O/r5_pmc_insts_pass.log 2>&1 || { tail -n 20 $O/r5_pmc_insts_pass.log; exit 1; }
rm -rf $O/r5_pmc_FETCH_SIZE $O/r5_pmc_WRITE_SIZE $O/r5_pmc_SQ $O/r5_pmc_INSTS
step "parity margins"; timeout -k 10 600 python -m pytest tests/test_gpu_philox_replay.py tests/test_gpu_configs.py -m gpu -q -s -k "metric" > $O/r5_parity.log 2>&1 || { tail -n 20 $O/r5_parity.log; exit 1; }
grep -E "vs float64|vs float32 oracle" $O/r5_parity.log > $O/r5_parity_margins.txt; cat $O/r5_parity_margins.txt | cut -c1-260
step "bench default again (traffic and issue now stamped)"; cp $O/r5_pmc_traffic.json profiles/r5_pmc_traffic.json; cp $O/r5_pmc_sq.json profiles/r5_pmc_sq.json
timeout -k 10 400 python bench.py > $O/r5_bench_default.json 2> $O/r5_bench_default.err || { tail -n 20 $O/r5_bench_default.err; exit 1; }
python -c "
import json;d=json.load(open('gpurun_out/r5_bench_default.json'));print('default (stamped): %.1f scenes/s, sustained %.1f, frac %.3f, traffic %s (%s), issue %s'%(d['value'],d['sustained']['value_this_rank'],d['roofline']['frac'],d['roofline']['traffic'],d['roofline'].get('traffic_source'),str(d['roofline'].get('issue'))[:120]))"
echo done
