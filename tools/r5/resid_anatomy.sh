cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in wtn; do echo "== $v"; VISRAG_HIP_LIB=$PWD/visrag_amd/libvisrag_hip_$v.so python tools/w_anatomy.py 32768,1152,1152,3 32768,1152,4352,3 2176,2304,2304,2 2176,2304,5760,2 2>&1 | tail -4 | cut -c1-330; done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_config1.py tests/test_gpu_config1xl.py -x -q 2>&1 | tail -4
python tools/ab_encode.py 10 2 2>/dev/null | tail -1 | cut -c1-400
