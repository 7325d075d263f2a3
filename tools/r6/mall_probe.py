"""Round 6 probe: does a launch that has just READ a cold weight matrix (so that it sits in the memory-side cache) take the
HBM latency out of the half-height GEMM that uses it next?  Trains of [read W_i ; GEMM(W_i)] against [read W_i] alone and
[GEMM(W_i)] alone, weights cycling through 40 buffers."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.gpu_util import P
from visrag_amd import _lib
lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
T, POOL = 2176, 40
for name, N, K in (("o", 2304, 2304), ("down", 2304, 5760)):
    A = torch.randn((2304, K), device="cuda").to(torch.bfloat16)
    Ws = [(torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16) for _ in range(POOL)]
    Wi = [w.view(torch.int32).view(-1) for w in Ws]
    out = torch.zeros((2304, N), device="cuda", dtype=torch.float32)
    sink = torch.zeros((), device="cuda", dtype=torch.int64)

    def gemm(i):
        _lib.check(lib.vr_op_gemm(0, P(A), K, P(Ws[i % POOL]), K, T, N, K, 3, None, P(out), 0.0, P(out), N, None, None, 0, 14, s))

    def read(i):
        torch.sum(Wi[i % POOL], dim=(0,), dtype=torch.int64, out=sink)

    def train(fn, n=3 * POOL):
        fn(0); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t_g = train(gemm)
    t_r = train(read)
    t_rg = train(lambda i: (read(i), gemm(i)))
    t_r1g = train(lambda i: (read(i + 1), gemm(i)))     # the read one launch AHEAD of its GEMM
    print(json.dumps({"case": name, "gemm_cold_us": round(t_g, 1), "read_us": round(t_r, 1), "read_then_gemm_us": round(t_rg, 1),
                      "gemm_after_read_us": round(t_rg - t_r, 1), "read_next_then_gemm_us": round(t_r1g, 1)}), flush=True)
