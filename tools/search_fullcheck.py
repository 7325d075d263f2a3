"""BASELINE-size exactness check: 1000 queries x 100k rows x 2304, top-10 ids vs a torch fp32 brute force for EVERY query
(mismatches are reported with the fp32 score gap: only rounding-level ties are tolerated), for three seeds, k = 10 and 100."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visrag_amd.engine import HipIndex
nd, dim, nq = 100_000, 2304, 1000
for seed in (0, 1, 2):
    g = torch.Generator(device="cuda").manual_seed(seed)
    C = torch.randn((nd, dim), generator=g, device="cuda"); C = C / C.norm(dim=1, keepdim=True)
    Q = torch.randn((nq, dim), generator=g, device="cuda"); Q = Q / Q.norm(dim=1, keepdim=True)
    ix = HipIndex(dim, nd); ix.add(C)
    ref = Q.double() @ C.double().T
    for k in (10, 100):
        sc, ids = ix.search(Q, k)
        rv, ri = torch.topk(ref, k, dim=1)
        bad = (ids != ri)
        nbad = int(bad.any(dim=1).sum())
        worst = 0.0
        if nbad:
            got_true = torch.gather(ref, 1, ids.clamp(min=0))
            worst = float((rv - got_true).abs()[bad].max())
        print(json.dumps({"seed": seed, "k": k, "queries_with_any_difference": nbad, "max_fp64_score_gap_at_differences": worst,
                          "max_score_err": float((sc.double() - torch.gather(ref, 1, ids)).abs().max())}))
    ix.close()
