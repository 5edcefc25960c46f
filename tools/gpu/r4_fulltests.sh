#!/bin/bash
set -o pipefail
O=gpurun_out; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/r4_gpu_tests.log 2>&1; rc=$?
tail -n 15 $O/r4_gpu_tests.log
exit $rc
