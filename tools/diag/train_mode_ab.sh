#!/bin/bash
# A/B of a library mode on the training step, ONE box, alternating: bash tools/diag/train_mode_ab.sh <rounds> <key=value> [train_bench flags]
set -o pipefail
ROUNDS=$1; MODE=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  echo -n "default      "; timeout -k 10 300 python tools/train_bench.py "$@" 2>/dev/null | grep "train step" || exit 1
  echo -n "$MODE "; timeout -k 10 300 python tools/train_bench.py "$@" --mode $MODE 2>/dev/null | grep "train step" || exit 1
done
