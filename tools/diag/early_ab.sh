#!/bin/bash
# A/B on one box: residual / second-source requests BEHIND the matrix phase's first weight loads (default build) against IN FRONT of the phase
# (as until round 5: build with GENCOMM_EXTRA_FLAGS=-DHC_EARLY_IN_FRONT GENCOMM_HIP_LIB=gencomm_amd/libgencomm_front.so), alternating
set -o pipefail
bash tools/gpu/r5_call3.sh front=gencomm_amd/libgencomm_front.so new2= front2=gencomm_amd/libgencomm_front.so 2>&1 | grep -v "shares" | tee gpurun_out/r5_early_ab.txt
