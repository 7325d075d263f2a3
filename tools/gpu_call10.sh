#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/c10; mkdir -p $O
bash tools/pmc_kernel.sh $O gemm python $PWD/tools/gemm_only.py gemm > /dev/null 2>&1
grep -v "^W" $O/pmc_gemm.txt | awk '{k=$2; for(i=3;i<=NF;i++) if ($i ~ /^SQ_/) {print k, $(i), $(i+1)}}' | sort | grep -E 'gemm256|gemm192' | head -150
