"""What cold operands cost a GEMM launch (round 6): the decoder's GEMM shapes with the weights cycling through a pool of 40
buffers (every launch reads its W from HBM, as in the model: 40 layers x 122 MB) against the same launches on ONE buffer
(L2 / memory-side-cache warm), HIP-event timed over the whole train of launches.
    python tools/cold_gemm.py            # shapes: qkv+RoPE-free bf16, gate/up SwiGLU, o / down residual on variants 13 / 14
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import P
from visrag_amd import _lib
lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
T = 2176
POOL = 40
cases = [("qkv bf16", 6912, 2304, 0, 12), ("gate/up swiglu", 11520, 2304, 4, 12), ("gate/up swiglu 128x256", 11520, 2304, 4, 15),
         ("o resid 256x192 nosplit", 2304, 2304, 3, 13), ("o resid 128x192", 2304, 2304, 3, 14),
         ("down resid 256x192 nosplit", 2304, 5760, 3, 13), ("down resid 128x192", 2304, 5760, 3, 14)]
for name, N, K, epi, variant in cases:
    A = [torch.randn((2304, K), device="cuda").to(torch.bfloat16) for _ in range(2)]
    Ws = [(torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16) for _ in range(POOL)]
    ocols = N // 2 if epi == 4 else N
    out = torch.zeros((2304, ocols), device="cuda", dtype=torch.float32 if epi == 3 else torch.bfloat16)
    resid = out if epi == 3 else None

    def run(n, cold_w, cold_a):
        for i in range(n):
            W = Ws[i % POOL] if cold_w else Ws[0]
            Ai = A[i % 2] if cold_a else A[0]
            _lib.check(lib.vr_op_gemm(0, P(Ai), K, P(W), K, T, N, K, epi, None, P(resid), 0.0 if epi == 3 else 1.0, P(out), ocols, None, None, 0, variant, s))
    res = {"case": name, "N": N, "K": K, "variant": variant, "W_MB": round(N * K * 2 / 1e6, 1)}
    for label, cw, ca in (("warm", False, False), ("cold_w", True, False)):
        run(POOL, cw, ca)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(3 * POOL, cw, ca); e1.record(); torch.cuda.synchronize()
        res[label + "_us"] = round(e0.elapsed_time(e1) / (3 * POOL) * 1e3, 1)
    res["tf_cold"] = round(2.0 * T * N * K / res["cold_w_us"] / 1e6, 0)
    print(json.dumps(res), flush=True)
