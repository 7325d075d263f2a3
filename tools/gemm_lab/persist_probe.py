"""Why is the persistent 256^2 GEMM (variant 10) +5 % alone but -20 % inside the model?  Time the GEMM with HIP events
(a) back to back, (b) with its A operand rewritten by the LayerNorm kernel before every launch (as in a ViT block)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.gpu_util import P
from visrag_amd import _lib
lib = _lib.load()
M, N, K = 32768, 3456, 1152
x = torch.randn((M, K), device="cuda")
w = torch.ones(K, device="cuda"); b = torch.zeros(K, device="cuda")
A = torch.empty((M, K), device="cuda", dtype=torch.bfloat16)
W = (torch.randn((3584, K), device="cuda") * 0.05).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
out = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16)
s = torch.cuda.current_stream().cuda_stream
def ln():
    _lib.check(lib.vr_op_norm(0, 0, P(x), M, K, P(w), P(b), 1e-6, P(A), K, s))
import ctypes as C
lab = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvisrag_gemm_lab.so")).vr_lab_gemm
lab.restype = C.c_int; lab.argtypes = lib.vr_op_gemm.argtypes
def gemm(v):
    f = lib.vr_op_gemm if v in (0, 3, 7, 9) else lab
    _lib.check(f(0, P(A), K, P(W), K, M, N, K, 0, P(bias), None, 1.0, P(out), N, None, None, 0, v, s))
ln(); torch.cuda.synchronize()
for v in (9, 10):
    for mode in ("back-to-back", "after LayerNorm"):
        tot = 0.0
        for it in range(12):
            if mode != "back-to-back": ln()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gemm(v); e1.record(); torch.cuda.synchronize()
            if it >= 2: tot += e0.elapsed_time(e1)
        ms = tot / 10
        print(json.dumps({"variant": v, "mode": mode, "ms": round(ms, 4), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}))
# (c) sustained: 300 x [LayerNorm, GEMM] without host syncs (clocks / power settle like in the model)
for v in (9, 10, 9, 10):
    for _ in range(20): ln(); gemm(v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): ln(); gemm(v)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"variant": v, "mode": "sustained LN+GEMM", "ms_per_pair": round(e0.elapsed_time(e1) / 300, 4)}))
