"""Run one GEMM (ViT qkv shape) back to back for N seconds; prints the sustained TFLOP/s.  For tools/gemm_power_probe.sh."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import P
from visrag_amd import _lib
lib = _lib.load()
v = sys.argv[1]; secs = float(sys.argv[2])
M, N, K = 32768, 3456, 1152
A = torch.randn((M, K), device="cuda").to(torch.bfloat16)
W = (torch.randn((3584, K), device="cuda") * 0.05).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
out = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16)
s = torch.cuda.current_stream().cuda_stream
Wt = W[:N].t()
def run():
    if v == "vendor": torch.matmul(A, Wt, out=out)
    else: _lib.check(lib.vr_op_gemm(0, P(A), K, P(W), K, M, N, K, 0, P(bias), None, 1.0, P(out), N, None, None, 0, int(v), s))
for _ in range(10): run()
torch.cuda.synchronize()
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    for _ in range(200): run()
    torch.cuda.synchronize(); n += 200
dt = time.time() - t0
print(json.dumps({"variant": v, "launches": n, "tflops": round(2.0 * M * N * K * n / dt / 1e12, 1)}))
