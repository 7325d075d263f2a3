#!/bin/bash
# Round 6, decoder o / down on half-height one-wave tiles (gemm128w.hip): op tests, then the in-model A/B on ONE box
# (product = 128w o/down + prefetch workgroups; _nopf = without the prefetch; _old = the split-K planes of rounds 2-5).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6/ab_dec4; mkdir -p $O
rocm-smi --showuniqueid --showclocks --showpower > $O/box.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm" > $O/ops.log 2>&1; tail -3 $O/ops.log
AB_PROFILE=2 bash tools/ab_libs.sh $O/dec 2 visrag_amd/libvisrag_hip.so visrag_amd/libvisrag_hip_nopf.so visrag_amd/libvisrag_hip_old.so
AB_PROFILE=1 bash tools/ab_libs.sh $O/all 3 visrag_amd/libvisrag_hip.so visrag_amd/libvisrag_hip_nopf.so visrag_amd/libvisrag_hip_old.so
