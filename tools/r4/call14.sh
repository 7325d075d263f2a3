#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c14; mkdir -p $O
for L in libvisrag_hip.so; do
  echo $L
  VISRAG_HIP_LIB=$PWD/visrag_amd/$L timeout 300 python tools/search_templated.py 1000 10 2>&1 | tail -1 | tee -a $O/templated.txt
  VISRAG_HIP_LIB=$PWD/visrag_amd/$L timeout 300 python tools/search_templated.py 100 10 2>&1 | tail -1 | tee -a $O/templated.txt
  VISRAG_HIP_LIB=$PWD/visrag_amd/$L timeout 300 python tools/search_templated.py 1 10 2>&1 | tail -1 | tee -a $O/templated.txt
done
timeout 600 python tools/hunt_band.py 40 2>&1 | tail -3
