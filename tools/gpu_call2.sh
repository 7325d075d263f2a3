#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/c2; mkdir -p $O
for L in _p1 ""; do
  VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip$L.so bash tools/pmc_kernel.sh $O/attn$L attention python $PWD/tools/gemm_only.py attn > /dev/null 2>&1
done
cat $O/attn_p1/pmc_attention.txt; cat $O/attn/pmc_attention.txt
