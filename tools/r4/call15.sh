#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_search.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
bash tools/r4/call14.sh
