#!/bin/bash
# What the big sweep's filter epilogue costs: the product library against two timing builds of search256w.hip
# (bash tools/variant.sh swd1 search256w.hip -DSW_DBG=1: no filter; swd2 -DSW_DBG=2: compares + ballots, no stores / counters).
# Their RESULTS are wrong (empty lists: every query takes the fallback passes) — only the `sweep` stage is read.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6/sweep_anatomy; mkdir -p $O
for L in "" _swd1 _swd2 ""; do
  echo "lib$L" >> $O/log.txt
  VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip$L.so timeout 300 python tools/search_diag.py 100000 2304 1000 x 2>/dev/null >> $O/log.txt
done
cat $O/log.txt | cut -c1-200
