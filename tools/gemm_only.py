import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import bench_gemm, bench_attn
import json
which = sys.argv[1] if len(sys.argv) > 1 else "gemm"
if which == "gemm":
    print(json.dumps(bench_gemm(32768, 4352, 1152, 1, 3)))     # fc1 + GELU   (256^2 kernel)
    print(json.dumps(bench_gemm(32768, 1152, 4352, 3, 3)))     # fc2 + resid  (256x192 kernel)
    print(json.dumps(bench_gemm(32768, 3456, 1152, 0, 3)))     # qkv
    print(json.dumps(bench_gemm(8192, 8192, 8192, 0, 3)))
elif which == "occ":
    for v in (0, 0x100):
        print(json.dumps(bench_gemm(100096, 1024, 2304, 0, v)))
else:
    print(json.dumps(bench_attn()))
