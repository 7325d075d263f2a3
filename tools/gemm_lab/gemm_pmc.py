import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.microbench import bench_gemm
v = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if v != 99:
    print(json.dumps(bench_gemm(8192, 8192, 8192, 0, v)))
    print(json.dumps(bench_gemm(32768, 3456, 1152, 0, v)))

if v == 99:
    for vv in (23, 26):
        print(json.dumps(bench_gemm(8192, 8192, 8192, 0, vv)))
        print(json.dumps(bench_gemm(32768, 3456, 1152, 0, vv)))
