#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 500 python tools/diag/shell_train_host.py > $O/r4c2_shell_host.txt 2>&1 || { tail -n 30 $O/r4c2_shell_host.txt; exit 1; }
head -n 12 $O/r4c2_shell_host.txt
echo done
