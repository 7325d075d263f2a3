#!/bin/bash
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4c11; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in libvisrag_hip_noas.so libvisrag_hip.so; do
  rm -rf /tmp/rp_$L
  VISRAG_HIP_LIB=$GRAFT_REPO_ROOT/visrag_amd/$L rocprofv3 --kernel-trace --stats -d /tmp/rp_$L -o e -- python $GRAFT_REPO_ROOT/tools/encode_only.py 3 > /tmp/rp.log 2>&1
  python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find /tmp/rp_$L -name '*.db' | head -1) $O/trace_$L.txt
  grep -i "attention" $O/trace_$L.txt | cut -c1-130
done
