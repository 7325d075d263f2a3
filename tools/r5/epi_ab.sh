cd "${GRAFT_REPO_ROOT:-/root/repo}"
VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip_wtn.so python tools/w_anatomy.py 2176,11520,2304,4 2>&1 | tail -1 | cut -c1-330
VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip_base.so python tools/gemm_only.py 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_config1.py tests/test_gpu_config1xl.py tests/test_gpu_encode.py tests/test_gpu_evisrag.py -x -q 2>&1 | tail -4
bash tools/ab_libs.sh gpurun_out/ab_epi2 3 visrag_amd/libvisrag_hip_base.so visrag_amd/libvisrag_hip.so | cut -c1-330
