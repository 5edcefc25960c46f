#!/bin/bash
# HBM-side traffic of the benchmark's kernels: two separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit
# into one pass on gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots") of the SAME bench.py command, one scene batch in
# flight, then tools/pmc_to_json.py -> profiles/${R}_pmc_traffic.json (read by bench.py for roofline.traffic).
#   bash tools/pmc_pass.sh [workload]        (on the GPU box; outputs under gpurun_out/)
set -o pipefail
export R=${ROUND:-r5}   # prefix of the output files (profiles/<round>_pmc_*.json)
WL=${1:-metric}
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/${R}_pmc_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${R}_pmc_$c -o pmc -- python3 bench.py --workload $WL --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-exact --no-timer > gpurun_out/${R}_pmc_$c.log 2>&1 || { tail -n 20 gpurun_out/${R}_pmc_$c.log; exit 1; }
done
python tools/pmc_to_json.py $WL 4 gpurun_out/${R}_pmc_FETCH_SIZE gpurun_out/${R}_pmc_WRITE_SIZE > gpurun_out/${R}_pmc_traffic.json && cat gpurun_out/${R}_pmc_traffic.json | head -c 1500
