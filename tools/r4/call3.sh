#!/bin/bash
# round 4, call 3: depth of the residual prefetch ring in the RESID epilogue of gemm256w.hip, in-model A/B on one box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c3; mkdir -p $O
bash tools/ab_libs.sh $O/ab 2 visrag_amd/libvisrag_hip_rd1.so visrag_amd/libvisrag_hip_rd2.so visrag_amd/libvisrag_hip_rd3.so visrag_amd/libvisrag_hip_rd4.so visrag_amd/libvisrag_hip_rd5.so visrag_amd/libvisrag_hip_rd6.so 2>&1 | tee $O/ab_summary.txt
python - <<'PY'
import json
for l in open("gpurun_out/r4c3/ab/ab.log"):
    j = json.loads(l); print(j["lib"].split("/")[-1], j["best"]["ms_per_step"], j["checksum"])
PY
