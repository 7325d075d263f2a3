#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/c7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "ln_fold" > $O/tests_ln.log 2>&1; tail -30 $O/tests_ln.log
timeout 900 python -m pytest tests/test_gpu_encode.py -q > $O/tests_enc.log 2>&1; grep -E "passed|failed|Error|cos|assert" $O/tests_enc.log | head -40
