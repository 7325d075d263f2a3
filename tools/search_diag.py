#!/usr/bin/env python
"""Per-stage times (HIP events inside vr_index_search) and certification counters of the fused search for several
batch sizes over a random unit-norm index:  python tools/search_diag.py [rows] [dim] [nq,nq,...] [certified-only]
SEARCH_DIAG_FRESH=1: every timed call gets DIFFERENT queries (24 sets in rotation).  With one query set repeated, the rows the
merge re-scores are the same every call and stay in the 256 MB memory-side cache when the sweep streams the index with the nt
policy: the merge then looks 12-20 us faster than it is for queries the index has not just seen."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visrag_amd.engine import HipIndex

nd = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 2304
g = torch.Generator(device="cuda").manual_seed(0)
C = torch.randn((nd, dim), generator=g, device="cuda"); C /= C.norm(dim=1, keepdim=True)
FRESH = os.environ.get("SEARCH_DIAG_FRESH") == "1"
NSET = 24 if FRESH else 1
Qall = torch.randn((NSET, 1000, dim), generator=g, device="cuda"); Qall /= Qall.norm(dim=2, keepdim=True)
ix = HipIndex(dim, nd); ix.add(C)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
NQS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 16, 64, 256, 1000]
for eps in ((None,) if len(sys.argv) > 4 else (None, -1.0)):
    ix.set_search_eps(eps)
    for nq in NQS:
        Qs = [Qall[i, :nq].contiguous() for i in range(NSET)]
        Q = Qs[0]
        for _ in range(3):
            ix.search(Q, 10)
        ix.search_stats(reset=True)
        e0.record()
        for it in range(20):
            ix.search(Qs[(it + 3) % NSET], 10)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        st = ix.search_stats(reset=True)
        ix.set_search_profile(True)
        for it in range(10):
            ix.search(Qs[(it + 13) % NSET], 10)
        pr = ix.get_search_profile(); ix.set_search_profile(False)
        print(f"eps={'default' if eps is None else 'off'} nq={nq:5d}  {ms*1e3:8.1f} us/search   stages(us): " +
              " ".join(f"{k}={v*1e3:.1f}" for k, v in pr.items() if k != "calls") + f"   per-20-calls: {st}", flush=True)
