"""ViT-shape attention (32 x 1024 tokens, 16 heads x 72) a few times through vr_op_attention — the microbench behind
tools/pmc_kernel.sh for this kernel (VISRAG_HIP_LIB picks a tagged build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.gpu_util import P
from visrag_amd import _lib
lib = _lib.load(os.environ.get("VISRAG_HIP_LIB") or None)
B, N, heads, hd = 32, 1024, 16, 72
W = heads * hd; ld = (3 * W + 127) // 128 * 128
qkv = torch.randn((B * N, ld), device="cuda").to(torch.bfloat16)
out = torch.zeros((B * N, (W + 127) // 128 * 128), dtype=torch.bfloat16, device="cuda")
cu = (torch.arange(B + 1, dtype=torch.int32) * N).cuda()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    _lib.check(lib.vr_op_attention(0, P(qkv), ld, qkv.data_ptr() + W * 2, ld, qkv.data_ptr() + 2 * W * 2, ld, P(out), out.stride(0), P(cu), P(cu),
                                   B, heads, hd, N, 0, 0, hd ** -0.5, None))
torch.cuda.synchronize()
