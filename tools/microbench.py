"""Micro-benchmarks of single kernels through the C ABI (HIP-event timed, random data)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tests.gpu_util import P  # noqa: E402
from visrag_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda:0"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def bench_attn(B=32, N=1024, heads=16, hd=72, causal=0):
    W = heads * hd
    ld = (3 * W + 127) // 128 * 128
    qkv = (torch.randn((B * N, ld), device=dev)).to(torch.bfloat16)
    out = torch.zeros((B * N, (W + 127) // 128 * 128), dtype=torch.bfloat16, device=dev)
    cu = (torch.arange(B + 1, dtype=torch.int32) * N).to(dev)
    s = torch.cuda.current_stream().cuda_stream

    def fn():
        _lib.check(lib.vr_op_attention(0, P(qkv), ld, qkv.data_ptr() + W * 2, ld, qkv.data_ptr() + 2 * W * 2, ld, P(out),
                                       out.stride(0), P(cu), P(cu), B, heads, hd, N, causal, 0, hd ** -0.5, s))
    ms = timeit(fn)
    fl = 4.0 * B * N * N * W * (0.5 if causal else 1.0)
    return {"op": f"attn B{B} N{N} h{heads} d{hd} c{causal}", "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}


def bench_gemm(M, N, K, epi=0, variant=0):
    r256 = lambda x: (x + 255) // 256 * 256
    A = torch.randn((r256(M), K), device=dev).to(torch.bfloat16)
    Wt = (torch.randn((r256(N), K), device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn((N,), device=dev)
    ocols = N // 2 if epi == 4 else N
    odt = torch.float32 if epi in (2, 3) else torch.bfloat16
    out = torch.zeros((M, ocols), dtype=odt, device=dev)
    resid = torch.zeros((M, ocols), dtype=torch.float32, device=dev) if epi == 3 else None
    s = torch.cuda.current_stream().cuda_stream

    if variant not in (0, 3, 7, 9, 12, 13):
        raise SystemExit(f"variant {variant} was a round-1 experiment (DESIGN.md ledger row 5); it is no longer built")
    call = lib.vr_op_gemm

    def fn():
        _lib.check(lib.vr_op_attention(0, P(qkv), ld, qkv.data_ptr() + W * 2, ld, qkv.data_ptr() + 2 * W * 2, ld, P(out),
                                       out.stride(0), P(cu), P(cu), B, heads, hd, N, causal, 0, hd ** -0.5, s))
    ms = timeit(fn)
    fl = 4.0 * B * N * N * W * (0.5 if causal else 1.0)
    return {"op": f"attn B{B} N{N} h{heads} d{hd} c{causal}", "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}


def bench_gemm(M, N, K, epi=0, variant=0):
    r256 = lambda x: (x + 255) // 256 * 256
    A = torch.randn((r256(M), K), device=dev).to(torch.bfloat16)
    Wt = (torch.randn((r256(N), K), device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn((N,), device=dev)
    ocols = N // 2 if epi == 4 else N
    odt = torch.float32 if epi in (2, 3) else torch.bfloat16
    out = torch.zeros((M, ocols), dtype=odt, device=dev)
    resid = torch.zeros((M, ocols), dtype=torch.float32, device=dev) if epi == 3 else None
    s = torch.cuda.current_stream().cuda_stream

    if variant in (0, 3, 7, 9, 12, 13):
        call = lib.vr_op_gemm
    else:           # round-1 experiment variants: tools/gemm_lab (python tools/gemm_lab/build.py)
        import ctypes as C
        lp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_lab", "libvisrag_gemm_lab.so")
        if not os.path.exists(lp):
            raise SystemExit(f"variant {variant} lives in the GEMM lab: build it with `python tools/gemm_lab/build.py`")
        call = C.CDLL(lp).vr_lab_gemm
        call.restype = C.c_int
        call.argtypes = lib.vr_op_gemm.argtypes

    def fn():
        _lib.check(call(0, P(A), K, P(Wt), K, M, N, K, epi, P(bias), P(resid), 1.0, P(out), ocols, None, None, 0,
                        variant, s))
    ms = timeit(fn)
    return {"op": f"gemm M{M} N{N} K{K} epi{epi} v{variant}", "ms": round(ms, 4),
            "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    res = []
    bench_gemm(8192, 8192, 8192, 0, 0)      # warm-up: clocks / caches settle before the first measurement
    if which in ("all", "attn"):
        res.append(bench_attn())
        res.append(bench_attn(B=32, N=68, heads=36, hd=64, causal=1))
        res.append(bench_attn(B=8, N=660, heads=36, hd=64, causal=1))
    if which in ("all", "gemm"):
        M = 32768
        for v in (9,):
            res.append(bench_gemm(M, 3456, 1152, 0, v))
            res.append(bench_gemm(M, 4352, 1152, 1, v))
            res.append(bench_gemm(M, 1152, 4352, 3, v))
            res.append(bench_gemm(M, 1152, 1152, 3, v))
        res.append(bench_gemm(M, 1152, 4352, 3, 7))
        res.append(bench_gemm(M, 1152, 1152, 3, 7))
        res.append(bench_gemm(M, 1152, 1152, 0, 7))
        res.append(bench_gemm(M, 1152, 1152, 0, 0))
        res.append(bench_gemm(2176, 6912, 2304, 0))
        res.append(bench_gemm(2176, 11520, 2304, 4))
        res.append(bench_gemm(2176, 2304, 5760, 3))
        for v in (9,):
            res.append(bench_gemm(2176, 6912, 2304, 0, v))
            res.append(bench_gemm(8192, 8192, 8192, 0, v))
            res.append(bench_gemm(4096, 4096, 4096, 0, v))
    for r in res:
        print(json.dumps(r), flush=True)
