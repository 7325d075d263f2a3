#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c10; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k attention > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
bash tools/ab_libs.sh $O/ab 3 visrag_amd/libvisrag_hip_noas.so visrag_amd/libvisrag_hip.so 2>&1 | tee $O/ab_summary.txt
