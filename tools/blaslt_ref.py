"""Yardstick only (never used by the product): what does the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) reach on
the model's shapes on this box?  Prints TFLOP/s next to the in-tree kernel."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.microbench import bench_gemm, timeit
bench_gemm(8192, 8192, 8192, 0, 9)
for (M, N, K) in ((8192, 8192, 8192), (32768, 3456, 1152), (32768, 4352, 1152), (32768, 1152, 4352), (2176, 6912, 2304), (2176, 11520, 2304)):
    A = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    W = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    ms = timeit(lambda: torch.matmul(A, W.t(), out=out))
    mine = bench_gemm(M, N, K, 0, 3)
    print(json.dumps({"shape": [M, N, K], "vendor_tflops": round(2.0 * M * N * K / ms / 1e9, 1), "ours_tflops": mine["tflops"]}))
