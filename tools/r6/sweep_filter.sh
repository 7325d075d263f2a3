#!/bin/bash
# The big sweep's mask filter (search256w.hip, round 6): the search tests, then stage times against the previous build
# (visrag_amd/libvisrag_hip_prev.so when present) on ONE box.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6/sweep_filter; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_search.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for L in "" _prev "" _prev; do
  [ -f visrag_amd/libvisrag_hip$L.so ] || continue
  echo "lib$L" >> $O/log.txt
  VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip$L.so timeout 300 python tools/search_diag.py 100000 2304 256,1000 x 2>/dev/null >> $O/log.txt
  VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip$L.so ND=12500 timeout 300 python tools/search_bench.py 1000 2>/dev/null >> $O/log.txt
done
cat $O/log.txt | cut -c1-230
timeout 600 python tools/search_templated.py 1000 10 2>/dev/null | tail -1 | cut -c1-400
