#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/c4; mkdir -p $O
VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip_p3.so timeout 300 python -m pytest tests/test_gpu_ops.py -q -k attention > $O/tests_p3.log 2>&1; echo "rc=$?" >> $O/tests_p3.log
timeout 300 python tools/ab_attention.py visrag_amd/libvisrag_hip_p1.so visrag_amd/libvisrag_hip_p3.so > $O/ab_attn.log 2>&1
VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip_p3.so timeout 200 python tools/ab_encode.py 10 3 >> $O/ab_enc.log 2>&1
VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip_p3.so bash tools/pmc_kernel.sh $O/attn_p3 attention python $PWD/tools/gemm_only.py attn > /dev/null 2>&1
tail -4 $O/tests_p3.log; cat $O/ab_attn.log | tail -12; tail -2 $O/ab_enc.log; cat $O/attn_p3/pmc_attention.txt | grep -v "^W" | awk '{print $5, $6}'
