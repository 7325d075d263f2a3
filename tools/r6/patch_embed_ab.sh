#!/bin/bash
# patch_embed with its two LDS tables (round 6) against the previous build: encode tests, then the in-model A/B (the checksum of
# the embeddings must not move: the conversion table holds the SAME floats) and the kernel's own time under rocprofv3
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6/patch_embed; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_config1.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
bash tools/ab_libs.sh $O 2 visrag_amd/libvisrag_hip.so visrag_amd/libvisrag_hip_prev.so | cut -c1-260
grep -o '"checksum": [0-9.]*' $O/ab.log | sort | uniq -c
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rpe && rocprofv3 --kernel-trace --stats -d /tmp/rpe -o e -- python $GRAFT_REPO_ROOT/tools/encode_only.py 2 > /tmp/rpe.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find /tmp/rpe -name '*.db' | head -1) $GRAFT_REPO_ROOT/$O/trace.txt; grep -i "patch_embed" $GRAFT_REPO_ROOT/$O/trace.txt | cut -c1-120
