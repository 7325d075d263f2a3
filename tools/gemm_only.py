import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import bench_gemm, bench_attn
import json
which = sys.argv[1] if len(sys.argv) > 1 else "gemm"
if which == "gemm":
    print(json.dumps(bench_gemm(32768, 4352, 1152, 1)))
    print(json.dumps(bench_gemm(32768, 1152, 4352, 3)))
    print(json.dumps(bench_gemm(8192, 8192, 8192, 0)))
else:
    print(json.dumps(bench_attn()))
