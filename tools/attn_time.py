"""time vr_op_attention on the ViT shape for every library on the command line (interleaved rounds)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import P
from visrag_amd import _lib
libs = [(os.path.basename(p), _lib.load(p)) for p in sys.argv[1:]]
dev = "cuda:0"
B, N, heads, hd = 32, 1024, 16, 72
W = heads * hd
ld = (3 * W + 127) // 128 * 128
qkv = torch.randn((B * N, ld), device=dev).to(torch.bfloat16)
out = torch.zeros((B * N, (W + 127) // 128 * 128), dtype=torch.bfloat16, device=dev)
cu = (torch.arange(B + 1, dtype=torch.int32) * N).to(dev)
def run(lib):
    _lib.check(lib.vr_op_attention(0, P(qkv), ld, qkv.data_ptr() + W * 2, ld, qkv.data_ptr() + 2 * W * 2, ld, P(out), out.stride(0), P(cu), P(cu), B, heads, hd, N, 0, 0, hd ** -0.5, None))
times = {n: [] for n, _ in libs}
for _ in range(4):
    for name, lib in libs:
        for _ in range(3): run(lib)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(lib)
        e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 20)
fl = 4.0 * B * N * N * W
for name, ts in times.items():
    ts = sorted(ts)
    print(f"{name:34s} min {ts[0]*1000:7.1f} us  med {ts[len(ts)//2]*1000:7.1f} us  {fl/ts[0]/1e9:7.1f} TF", flush=True)
