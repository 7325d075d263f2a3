#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/c5; mkdir -p $O
for L in p3 p4; do
VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip_$L.so timeout 300 python -m pytest tests/test_gpu_ops.py -q -k attention > $O/tests_$L.log 2>&1; echo "rc=$?" >> $O/tests_$L.log; tail -3 $O/tests_$L.log
done
timeout 300 python tools/ab_attention.py visrag_amd/libvisrag_hip_p1.so visrag_amd/libvisrag_hip_p3.so visrag_amd/libvisrag_hip_p4.so > $O/ab_attn.log 2>&1
cat $O/ab_attn.log | tail -12
