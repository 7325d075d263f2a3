"""tile anatomy of attention_w.hip from an AW_TIMING build: python tools/attn_anatomy.py <lib>"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.gpu_util import P
from visrag_amd import _lib
lib = _lib.load(sys.argv[1])
raw = C.CDLL(os.path.abspath(sys.argv[1]))
B, N, heads, hd = 32, 1024, 16, 72
W = heads * hd; ld = (3 * W + 127) // 128 * 128
qkv = torch.randn((B * N, ld), device="cuda").to(torch.bfloat16)
out = torch.zeros((B * N, (W + 127) // 128 * 128), dtype=torch.bfloat16, device="cuda")
cu = (torch.arange(B + 1, dtype=torch.int32) * N).cuda()
def run():
    _lib.check(lib.vr_op_attention(0, P(qkv), ld, qkv.data_ptr() + W * 2, ld, qkv.data_ptr() + 2 * W * 2, ld, P(out), out.stride(0), P(cu), P(cu), B, heads, hd, N, 0, 0, hd ** -0.5, None))
for _ in range(5): run()
torch.cuda.synchronize()
units = B * heads * 4
NWV = int(os.environ.get("AW_NW", "8")); buf = np.zeros((units, NWV, 64), dtype=np.uint64)
assert raw.vr_dbg_attn_timing_copy(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes)) == 0
t = buf.astype(np.int64)
tot = t[:, :, 3] - t[:, :, 0]; pro = t[:, :, 1] - t[:, :, 0]; loop = t[:, :, 2] - t[:, :, 1]; epi = t[:, :, 3] - t[:, :, 2]
wait = t[:, :, 4]; nb = t[:, :, 5]
tick = 1.0   # s_memtime ticks = shader cycles
print("s_memtime ticks (100 MHz):  per unit (mean over units, waves)")
print(f"  total {tot.mean()*tick:7.2f} cyc   prologue {pro.mean()*tick:6.2f}   loop {loop.mean()*tick:6.2f}   tail+epilogue {epi.mean()*tick:6.2f}")
print(f"  barriers {nb.mean():.1f}, wait+barrier total {wait.mean()*tick:6.2f} cyc = {wait.mean()/np.maximum(nb.mean(),1)*tick:5.3f} cyc each; per tile {loop.mean()/16*tick:5.3f} us")
st = t[:, :, 8:8 + 16]
d = np.diff(st, axis=2)
print("  barrier-to-barrier (us), tiles 1..15:", np.round(d.mean((0, 1)) * tick, 3))
# per-wave skew: max - min of the total over the 4 waves
print(f"  unit span over the grid: first start {t[:,:,0].min()}, last end {t[:,:,3].max()}, span {(t[:,:,3].max()-t[:,:,0].min())*tick:7.1f} us")
