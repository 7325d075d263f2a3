"""How do isolated GEMM numbers carry over to a stream of kernels?  For the in-tree 256^2 kernel (variant 9),
the one-wave kernel (12) and the vendor GEMM (torch.matmul, yardstick only):
(a) GEMM launches separated by host syncs, (b) 300 GEMMs back to back, (c) 300 x [LayerNorm, GEMM] back to back
(the ViT pattern: the A operand is rewritten before every launch); (c) minus the LayerNorm-only stream is the
GEMM's in-stream cost."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tests.gpu_util import P  # noqa: E402
from visrag_amd import _lib  # noqa: E402

lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream


def ev():
    return torch.cuda.Event(enable_timing=True)


def stream_ms(fn, n=300, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (M, N, K) in ((32768, 3456, 1152), (32768, 4352, 1152)):
    x = torch.randn((M, K), device="cuda")
    w = torch.ones(K, device="cuda"); b = torch.zeros(K, device="cuda")
    A = torch.empty((M, K), device="cuda", dtype=torch.bfloat16)
    Np = (N + 255) // 256 * 256
    W = (torch.randn((Np, K), device="cuda") * 0.05).to(torch.bfloat16)
    Wt = W[:N].t()
    bias = torch.randn(N, device="cuda")
    out = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16)

    def ln():
        _lib.check(lib.vr_op_norm(0, 0, P(x), M, K, P(w), P(b), 1e-6, P(A), K, s))

    def gemm(v):
        if v == "vendor":
            torch.matmul(A, Wt, out=out)
            return
        f = lib.vr_op_gemm
        _lib.check(f(0, P(A), K, P(W), K, M, N, K, 0, P(bias), None, 1.0, P(out), N, None, None, 0, v, s))

    ln(); torch.cuda.synchronize()
    ln_ms = stream_ms(ln)
    fl = 2.0 * M * N * K
    for v in [9, 12, "vendor"]:
        tot = 0.0
        for it in range(12):
            e0, e1 = ev(), ev()
            e0.record(); gemm(v); e1.record(); torch.cuda.synchronize()
            if it >= 2:
                tot += e0.elapsed_time(e1)
        iso = tot / 10
        b2b = stream_ms(lambda: gemm(v))
        pair = stream_ms(lambda: (ln(), gemm(v)))
        print(json.dumps({"shape": [M, N, K], "gemm": v, "synced_tflops": round(fl / iso / 1e9, 1),
                          "back_to_back_tflops": round(fl / b2b / 1e9, 1), "ln_only_ms": round(ln_ms, 4),
                          "ln_gemm_pair_ms": round(pair, 4), "in_stream_tflops": round(fl / (pair - ln_ms) / 1e9, 1)}))
