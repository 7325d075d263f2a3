"""Op-level A/B of attention builds: for every library path on the command line (tagged builds of
`python -m visrag_amd.build --tag ...`) time vr_op_attention on the ViT shape (32 x 1024 tokens, 16 heads x 72) and a
decoder shape, interleaved rounds, and check each build against a torch fp32 reference on a small case."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import P
from visrag_amd import _lib

paths = sys.argv[1:] or [_lib.LIB_PATH]
libs = [(os.path.basename(p), _lib.load(p)) for p in paths]
dev = "cuda:0"


def run(lib, qkv, out, cu, B, heads, hd, N, causal):
    W = heads * hd
    ld = qkv.stride(0)
    _lib.check(lib.vr_op_attention(0, P(qkv), ld, qkv.data_ptr() + W * 2, ld, qkv.data_ptr() + 2 * W * 2, ld, P(out),
                                   out.stride(0), P(cu), P(cu), B, heads, hd, N, causal, 0, hd ** -0.5, None))


def ref_check(lib, hd, lens, causal):
    heads = 2
    W = heads * hd
    T = sum(lens)
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn((T, 3 * W), generator=g).to(torch.bfloat16)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    d = qkv.to(dev); out = torch.zeros((T, W), dtype=torch.bfloat16, device=dev)
    run(lib, d, out, cu.to(dev), len(lens), heads, hd, max(lens), causal)
    torch.cuda.synchronize()
    worst = 0.0
    for b in range(len(lens)):
        lo, hi = int(cu[b]), int(cu[b + 1])
        for h in range(heads):
            q = qkv[lo:hi, h * hd:(h + 1) * hd].float(); k = qkv[lo:hi, W + h * hd:W + (h + 1) * hd].float()
            v = qkv[lo:hi, 2 * W + h * hd:2 * W + (h + 1) * hd].float()
            s = q @ k.T * hd ** -0.5
            if causal:
                s = s + torch.full(s.shape, float("-inf")).triu(1)
            r = torch.softmax(s, -1) @ v
            worst = max(worst, float((out[lo:hi, h * hd:(h + 1) * hd].float().cpu() - r).abs().max()))
    return worst


def bench(B, N, heads, hd, causal, rounds=5, iters=10):
    W = heads * hd
    ld = (3 * W + 127) // 128 * 128
    qkv = torch.randn((B * N, ld), device=dev).to(torch.bfloat16)
    out = torch.zeros((B * N, (W + 127) // 128 * 128), dtype=torch.bfloat16, device=dev)
    cu = (torch.arange(B + 1, dtype=torch.int32) * N).to(dev)
    times = {n: [] for n, _ in libs}
    for _ in range(rounds):
        for name, lib in libs:
            for _ in range(2):
                run(lib, qkv, out, cu, B, heads, hd, N, causal)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                run(lib, qkv, out, cu, B, heads, hd, N, causal)
            e1.record(); torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / iters)
    fl = 4.0 * B * N * N * W * (0.5 if causal else 1.0)
    for name, ts in times.items():
        ts = sorted(ts)
        print(json.dumps({"lib": name, "shape": f"B{B} N{N} h{heads} d{hd} c{causal}", "ms_min": round(ts[0], 4),
                          "ms_med": round(ts[len(ts) // 2], 4), "tflops_at_min": round(fl / ts[0] / 1e9, 1)}), flush=True)


for name, lib in libs:
    print(json.dumps({"lib": name, "max_abs_err": {"hd72": ref_check(lib, 72, [1026, 60], 0), "hd64_causal": ref_check(lib, 64, [68, 13, 130, 1, 700], 1),
                                                    "hd72_one_tile": ref_check(lib, 72, [64], 0), "hd72_129": ref_check(lib, 72, [129, 200], 0)}}), flush=True)
bench(32, 1024, 16, 72, 0)
bench(32, 68, 36, 64, 1, iters=20)
bench(16, 660, 36, 64, 1, iters=10)
