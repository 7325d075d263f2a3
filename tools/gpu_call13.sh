#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests -m gpu -x -q -k "encode or config1 or ops" 2>&1 | tail -3
bash tools/ab_libs.sh gpurun_out/c13 2 visrag_amd/libvisrag_hip_base.so visrag_amd/libvisrag_hip.so
