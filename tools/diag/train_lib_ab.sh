#!/bin/bash
# A/B of diagnostic library builds on the training step, ONE box, alternating: bash tools/diag/train_lib_ab.sh <rounds> name=path.so [...] -- [train_bench flags]
# (a build: GENCOMM_HIP_LIB=<path> GENCOMM_EXTRA_FLAGS="-D..." python -c "from gencomm_amd import _lib; _lib.build(force=True)")
set -o pipefail
ROUNDS=$1; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
for r in $(seq 1 $ROUNDS); do
  for kv in base= "${LIBS[@]}"; do
    name=${kv%%=*}; path=${kv#*=}
    if [ -n "$path" ]; then export GENCOMM_HIP_LIB=$PWD/$path; else unset GENCOMM_HIP_LIB; fi
    printf "%-10s " "$name"; timeout -k 10 300 python tools/train_bench.py "$@" 2>/dev/null | grep "train step" || exit 1
  done
done
