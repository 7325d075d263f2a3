#!/bin/bash
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r4c12; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encode.py tests/test_gpu_config1.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
bash tools/ab_libs.sh $O/ab 3 visrag_amd/libvisrag_hip_noas.so visrag_amd/libvisrag_hip.so 2>&1 | tee $O/ab_summary.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rpx; rocprofv3 --kernel-trace --stats -d /tmp/rpx -o e -- python $GRAFT_REPO_ROOT/tools/encode_only.py 3 > /tmp/rp.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find /tmp/rpx -name '*.db' | head -1) $O/trace.txt; grep -i "attention" $O/trace.txt | cut -c1-130
