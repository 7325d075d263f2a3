cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_config1.py tests/test_gpu_config1xl.py tests/test_gpu_encode.py -x -q 2>&1 | tail -4
bash tools/ab_libs.sh gpurun_out/ab_pairs 3 visrag_amd/libvisrag_hip_base.so visrag_amd/libvisrag_hip.so | cut -c1-330
