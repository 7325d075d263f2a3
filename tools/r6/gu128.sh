#!/bin/bash
# decoder gate / up on 128 x 256 half-height tiles (-DVR_DEC_GU_128W=1: 765 tiles = 2.99 rounds) against the 256 x 256 default
# (405 tiles = 1.58 rounds): op tests, then the in-model A/B on ONE box (decoder phases split: AB_PROFILE=2)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6/gu128; mkdir -p $O
VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip_gu128.so timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_config1.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
AB_PROFILE=2 bash tools/ab_libs.sh $O/dec 2 visrag_amd/libvisrag_hip.so visrag_amd/libvisrag_hip_gu128.so | cut -c1-330
AB_PROFILE=1 bash tools/ab_libs.sh $O/all 3 visrag_amd/libvisrag_hip.so visrag_amd/libvisrag_hip_gu128.so | cut -c1-200
grep -o '"checksum": [0-9.]*' $O/all/ab.log | sort | uniq -c
