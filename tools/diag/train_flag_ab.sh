#!/bin/bash
# A/B of a tools/train_bench.py flag on ONE box, alternating: bash tools/diag/train_flag_ab.sh <rounds> <flag> [train_bench flags]
set -o pipefail
ROUNDS=$1; FLAG=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  printf "%-12s " default; timeout -k 10 300 python tools/train_bench.py "$@" 2>/dev/null | grep "train step" | cut -c1-150 || exit 1
  printf "%-12s " "$FLAG"; timeout -k 10 300 python tools/train_bench.py "$@" $FLAG 2>/dev/null | grep "train step" | cut -c1-150 || exit 1
done
