#!/bin/bash
# Round 6: tile anatomy of the half-height one-wave GEMM (variant 14) at the decoder's o / down shapes, per raster group size,
# next to the 256-row tile without split K (variant 13).  Needs: python -m visrag_amd.build --tag wtm -DVR_W_TIMING
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6/h_anatomy; mkdir -p $O
export VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip_wtm.so
for gm in 1 2 4 6 17; do
  echo "GM $gm" >> $O/anatomy.log
  VR_H_GM=$gm python tools/w_anatomy.py 2176,2304,2304,3,14 2176,2304,5760,3,14 2>/dev/null >> $O/anatomy.log
done
echo "variant 13 (256 x 192, no split)" >> $O/anatomy.log
python tools/w_anatomy.py 2176,2304,2304,3,13 2176,2304,5760,3,13 2>/dev/null >> $O/anatomy.log
echo "variant 15 (128 x 256) / 12 (256 x 256), SwiGLU gate/up" >> $O/anatomy.log
VR_H_GM=4 python tools/w_anatomy.py 2176,11520,2304,4,15 2176,11520,2304,4,12 2>/dev/null >> $O/anatomy.log
cat $O/anatomy.log
