#!/bin/bash
# GPU call 1: full -m gpu suite on the product build + attention A/B (op level and in-model)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/c1; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 300 python tools/ab_attention.py visrag_amd/libvisrag_hip_p0.so visrag_amd/libvisrag_hip_p1.so visrag_amd/libvisrag_hip.so > $O/ab_attn.log 2>&1
for L in _p0 _p1 ""; do
  VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip$L.so timeout 200 python tools/ab_encode.py 10 3 >> $O/ab_enc.log 2>&1
done
tail -5 $O/tests.log; cat $O/ab_attn.log | tail -20; cat $O/ab_enc.log | tail -5
