import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visrag_amd.engine import HipIndex
nd, dim = int(os.environ.get("ND", "100000")), 2304      # ND=12500: one shard of the 8-GPU run
g = torch.Generator(device="cuda").manual_seed(0)
C = torch.randn((nd, dim), generator=g, device="cuda"); C = C / C.norm(dim=1, keepdim=True)
ix = HipIndex(dim, nd); ix.add(C)
nqs = [int(x) for x in sys.argv[1:]] or [1000, 128, 16, 1]
for nq in nqs:
    Q = torch.randn((nq, dim), generator=g, device="cuda"); Q = Q / Q.norm(dim=1, keepdim=True)
    for _ in range(3): ix.search(Q, 10)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): sc, ids = ix.search(Q, 10)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    rv, ri = torch.topk(Q[:16] @ C.T, 10, dim=1)
    ok = bool(torch.equal(ri, ids[:16]))
    print(json.dumps({"nq": nq, "ms": round(ms, 4), "qps": round(nq / ms * 1e3), "tflops": round(2.0 * nq * nd * dim / ms / 1e9, 1),
                      "index_GBps": round(nd * dim * 2 / ms / 1e6, 1), "ids_match_torch": ok}))
