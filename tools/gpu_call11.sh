#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "gemm" 2>&1 | tail -3
bash tools/ab_libs.sh gpurun_out/c11 3 visrag_amd/libvisrag_hip_base.so visrag_amd/libvisrag_hip.so
