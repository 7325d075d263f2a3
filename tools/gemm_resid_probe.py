"""How much of the N=1152 GEMMs (SigLIP proj / fc2) is their fp32 residual read-modify-write epilogue?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import bench_gemm
bench_gemm(8192, 8192, 8192, 0, 9)
for (M, N, K) in ((32768, 1152, 1152), (32768, 1152, 4352), (2176, 2304, 2304), (2176, 2304, 5760)):
    v = 7 if N == 1152 else 3
    print(json.dumps([bench_gemm(M, N, K, e, v) for e in (3, 2, 0)]))
