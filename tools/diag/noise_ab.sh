#!/bin/bash
# VALU budget of the sampler step's in-kernel noise, by subtraction: rocprofv3 durations of latent_step_h_kernel / q_sample_kernel under
# diagnostic builds (-DGC_NOISE_DIAG=bits, common.h) of the metric workload, one stream.  bash tools/diag/noise_ab.sh 1 2 4 7
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for v in base "$@"; do
  if [ $v = base ]; then unset GENCOMM_HIP_LIB; else export GENCOMM_HIP_LIB=$PWD/gencomm_amd/_build/variants/lib_nz$v.so; fi
  rm -rf $O/nzab_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/nzab_$v -o k -- python3 bench.py --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-exact --no-timer > $O/nzab_$v.log 2>&1 || { tail -n 20 $O/nzab_$v.log; exit 1; }
  T=$(find $O/nzab_$v -name "*kernel_trace.csv" | head -1)
  echo "== GC_NOISE_DIAG=$v"; python tools/trace_by_grid.py $T latent_step q_sample conv_out_h | cut -d, -f1-5 | head -5
  rm -rf $O/nzab_$v
done
