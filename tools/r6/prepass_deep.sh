#!/bin/bash
# The pre-pass's K loop with four LDS stages (search.hip: prepass_mainloop) against the two-stage build: tests, then stage times
# at 100 000 rows (owning pre-pass) and on an 8-way shard (12 500 rows: the plain one) on ONE box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6/prepass_deep; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_search.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
for L in "" _prev "" _prev; do
  echo "lib$L" >> $O/log.txt
  VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip$L.so timeout 300 python tools/search_diag.py 100000 2304 256,1000 x 2>/dev/null >> $O/log.txt
  VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip$L.so timeout 300 python tools/search_diag.py 12500 2304 1000 x 2>/dev/null >> $O/log.txt
done
cut -c1-170 $O/log.txt
