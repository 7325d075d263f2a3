#!/bin/bash
# tools/variant.sh <tag> <file.hip> [-D...]: libvisrag_hip_<tag>.so = the product objects with ONE file recompiled under extra defines
set -e
cd /root/repo/visrag_amd
TAG=$1; F=$2; shift 2
mkdir -p build_$TAG
FL=$(cd /root/repo && python -c "from visrag_amd.build import FILE_FLAGS; print(' '.join(FILE_FLAGS.get('$F', [])))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $FL "$@" -c csrc/$F -o build_$TAG/${F%.hip}.o
OBJS=""
for o in build/*.o; do b=$(basename $o); if [ "$b" == "${F%.hip}.o" ]; then OBJS="$OBJS build_$TAG/$b"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libvisrag_hip_$TAG.so $OBJS
echo libvisrag_hip_$TAG.so
