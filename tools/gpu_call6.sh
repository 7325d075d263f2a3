#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/c6; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -15 $O/tests.log
timeout 300 python tools/ab_attention.py visrag_amd/libvisrag_hip.so visrag_amd/libvisrag_hip_q1.so > $O/ab_attn.log 2>&1; tail -8 $O/ab_attn.log
timeout 200 python tools/ab_encode.py 10 3 > $O/ab_enc.log 2>&1; tail -1 $O/ab_enc.log

