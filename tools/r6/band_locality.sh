#!/bin/bash
# Does the band pass of the templated corpus want its flagged queries ordered by family?  (experiment: SORTQ=1 orders the
# QUERIES by family, libvisrag_hip_bxcd.so = consecutive slots of a round on one XCD: bash tools/variant.sh bxcd search_band.hip -DBAND_XCD=1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6/band_locality; mkdir -p $O
for L in "" _bxcd; do for SQ in 0 1; do
  echo "lib$L SORTQ=$SQ" >> $O/log.txt
  SORTQ=$SQ VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip$L.so timeout 600 python tools/search_templated.py 1000 10 2>/dev/null | tail -1 | cut -c1-330 >> $O/log.txt
done; done
cat $O/log.txt
