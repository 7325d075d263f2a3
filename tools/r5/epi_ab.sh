cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_config1.py tests/test_gpu_encode.py -x -q 2>&1 | tail -3
bash tools/ab_libs.sh gpurun_out/ab_rope 3 visrag_amd/libvisrag_hip_base.so visrag_amd/libvisrag_hip.so | cut -c1-330
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rpn && rocprofv3 --kernel-trace --stats -d /tmp/rpn -o e -- python $GRAFT_REPO_ROOT/tools/encode_only.py 2 > /tmp/rpn.log 2>&1; python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find /tmp/rpn -name '*.db' | head -1) /tmp/rpn.txt; grep "gemm256w_bf16_kernel<5" /tmp/rpn.txt | cut -c1-110
