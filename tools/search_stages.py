"""stage times of the search for shard sizes and query counts: ND=12500,25000,100000"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visrag_amd.engine import HipIndex
dim = 2304
g = torch.Generator(device="cuda").manual_seed(0)
for nd in [int(x) for x in os.environ.get("ND", "12500,25000,100000").split(",")]:
    C = torch.randn((nd, dim), generator=g, device="cuda"); C = C / C.norm(dim=1, keepdim=True)
    ix = HipIndex(dim, nd); ix.add(C)
    for nq in [1000, 16, 1]:
        Q = torch.randn((nq, dim), generator=g, device="cuda"); Q = Q / Q.norm(dim=1, keepdim=True)
        for _ in range(3): ix.search(Q, 10)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ix.search(Q, 10)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        ix.search_stats(reset=True); ix.set_search_profile(True)
        for _ in range(20): ix.search(Q, 10)
        st = ix.get_search_profile(); ix.set_search_profile(False)
        print(f"rows {nd:6d} nq {nq:4d}  {ms*1000:7.1f} us/search   stages(us): " + " ".join(f"{k}={v*1000:.1f}" for k, v in st.items() if k != "calls"), ix.search_stats(), flush=True)
    ix.close(); del C
