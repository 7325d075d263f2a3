#!/bin/bash
# Tile anatomy of the fp32-epilogue GEMMs (ViT proj / fc2 residual, decoder o / down split-K planes) and of qkv / fc1, from
# in-kernel timestamps: prologue, K-loop (with its shader clock), epilogue, gap between workgroups on a CU.
#   HERE (no GPU needed):  bash tools/variant.sh wtn gemm256w.hip -DVR_W_TIMING
#   on the GPU box:        bash tools/r5/resid_anatomy.sh          (DESIGN.md 5.2 / ledger 44 quote its output)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip_wtn.so python tools/w_anatomy.py 32768,1152,1152,3 32768,1152,4352,3 2176,2304,2304,2 2176,2304,5760,2 32768,3456,1152,0 32768,4352,1152,1 2>&1 | tail -6 | cut -c1-420
tools/probe_store 2>/dev/null | tail -21        # (hipcc --offload-arch=gfx950 -O3 -o tools/probe_store tools/probe_store.hip)
