#!/bin/bash
# certify_tail with two candidate rows per wave in flight (VR_RESCORE_PAIRS) against the one-row loop: stage times on ONE box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6/merge_pairs; mkdir -p $O
for L in "" _nopair "" _nopair; do
  echo "lib$L" >> $O/log.txt
  VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip$L.so SEARCH_DIAG_FRESH=1 timeout 300 python tools/search_diag.py 100000 2304 256,1000 x 2>/dev/null >> $O/log.txt
done
cut -c1-190 $O/log.txt
