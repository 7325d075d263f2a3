#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in "" _prev; do
  rm -rf /tmp/rpp$L; VISRAG_HIP_LIB=$R/visrag_amd/libvisrag_hip$L.so rocprofv3 --kernel-trace --stats -d /tmp/rpp$L -o s -- python $R/tools/search_bench.py 1000 > /tmp/rpp$L.log 2>&1
  python $R/tools/prof_summary.py $(find /tmp/rpp$L -name '*.db' | head -1) /tmp/rpp$L.txt; echo "lib$L"; grep "prepass\|thr_own\|sweep256w\|merge256" /tmp/rpp$L.txt | cut -c1-110
  rm -rf /tmp/rpq$L; ND=12500 VISRAG_HIP_LIB=$R/visrag_amd/libvisrag_hip$L.so rocprofv3 --kernel-trace --stats -d /tmp/rpq$L -o s -- python $R/tools/search_bench.py 1000 > /tmp/rpq$L.log 2>&1
  python $R/tools/prof_summary.py $(find /tmp/rpq$L -name '*.db' | head -1) /tmp/rpq$L.txt; echo "lib$L shard"; grep "prepass\|search_thr" /tmp/rpq$L.txt | cut -c1-110
done
