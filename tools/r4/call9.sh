#!/bin/bash
# round 4, call 9: one-wave-per-(sequence, head) attention for short sequences — parity, then in-model A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encode.py tests/test_gpu_config1.py tests/test_gpu_pipeline.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
bash tools/ab_libs.sh $O/ab 3 visrag_amd/libvisrag_hip_noas.so visrag_amd/libvisrag_hip.so 2>&1 | tee $O/ab_summary.txt
