#!/bin/bash
# Round 5, call 2: the staging "VALU diet" of the f16-pipe 8-channel kernels (GroupNorm + SiLU as z r in five instructions, every split
# residual as v_fma_mix_f32, no zero fills in front of the bf8 conversions, lane pointer + uniform offsets in the epilogue) -- full GPU
# suite, then A/B against the start-of-round library (gencomm_amd/libgencomm_base.so) alternating on this box.
set -o pipefail
O=gpurun_out; mkdir -p $O
echo "== gpu tests"; timeout -k 10 700 python -m pytest tests -m gpu -q -x > $O/r5c2_gpu_tests.log 2>&1; rc=$?; tail -n 8 $O/r5c2_gpu_tests.log; [ $rc -eq 0 ] || exit $rc
echo "== A/B"; bash tools/diag/lib_ab.sh r5c2_ab 3 old=gencomm_amd/libgencomm_base.so 2>&1 | tee $O/r5c2_ab.txt
echo done
