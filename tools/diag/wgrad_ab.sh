#!/bin/bash
# Where conv_wgrad_mfma_kernel's time goes: rocprofv3 kernel durations of the large training step under diagnostic builds (-DWG_DIAG=bits,
# unet_bwd_kernels.h), per launch shape.   bash tools/diag/wgrad_ab.sh v1 v2 ...   (variants = gencomm_amd/_build/variants/lib_wgd<v>.so)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for v in base "$@"; do
  if [ $v = base ]; then unset GENCOMM_HIP_LIB; else export GENCOMM_HIP_LIB=$PWD/gencomm_amd/_build/variants/lib_wgd$v.so; fi
  rm -rf $O/wgab_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/wgab_$v -o k -- python3 tools/train_bench.py --only large --no-optimizer > $O/wgab_$v.log 2>&1 || { tail -n 20 $O/wgab_$v.log; exit 1; }
  T=$(find $O/wgab_$v -name "*kernel_trace.csv" | head -1)
  echo "== WG_DIAG=$v"; python tools/trace_by_grid.py $T conv_wgrad_mfma | cut -d, -f1-5 | sed 's/"void gc::conv_wgrad_mfma_kernel/wgrad/' | head -8
  rm -rf $O/wgab_$v
done
