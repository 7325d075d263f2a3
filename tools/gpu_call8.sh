#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/c8; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -k "config1 or decoder_taps or checkpoint_dir or demo or two_ranks or pipeline" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -40 $O/tests.log
