#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_backward.py "tests/test_gpu_parity.py::test_training_gradients_match_reference" "tests/test_gpu_parity.py::test_train_mode_forward_matches_reference_train_branch" tests/test_shell.py -m gpu -q -s > gpurun_out/r2c6_pytest.log 2>&1
rc=$?
grep -n "worst relative\|passed\|failed\|Error\|error" gpurun_out/r2c6_pytest.log | head -40
tail -n 30 gpurun_out/r2c6_pytest.log
echo "pytest rc=$rc"
