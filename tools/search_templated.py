#!/usr/bin/env python
"""The certification's worst case as a workload for rocprofv3 / timing: a "templated" corpus — 100 families x 1000 pages laid out
CONTIGUOUSLY (a deck embedded page after page), pairwise cosine 0.97 .. 0.999 inside a family (random unit centres here;
bench.py perturbs model embeddings) — where nearly every query is flagged and redone by the band pass (search_band.hip).
    python tools/search_templated.py [queries=1000] [searches=10]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from visrag_amd.engine import HipIndex

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
E, n_fam, per = 2304, 100, 1000
g = torch.Generator(device="cuda").manual_seed(0)
centers = torch.randn((n_fam, E), generator=g, device="cuda"); centers /= centers.norm(dim=1, keepdim=True)
t_f = torch.logspace(np.log10(0.176), np.log10(0.0316), n_fam, device="cuda")
rows = torch.empty((n_fam * per, E), device="cuda")
for f in range(n_fam):
    r = centers[f][None, :] + t_f[f] * torch.randn((per, E), generator=g, device="cuda") / E ** 0.5
    rows[f * per:(f + 1) * per] = r / r.norm(dim=1, keepdim=True)
fam_q = torch.randint(0, n_fam, (nq,), generator=g, device="cuda")
if os.environ.get("SORTQ") == "1":          # (experiment: queries of a family adjacent — what a locality sort of the flagged queries would give)
    fam_q = fam_q.sort().values
Q = torch.randn((nq, E), generator=g, device="cuda") + 3.0 * centers[fam_q]
Q /= Q.norm(dim=1, keepdim=True)
ix = HipIndex(E, n_fam * per); ix.add(rows)
for _ in range(2):
    sc, ids = ix.search(Q, 10)
ix.search_stats(reset=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    sc, ids = ix.search(Q, 10)
e1.record(); torch.cuda.synchronize()
st = ix.search_stats()
ix.set_search_profile(True)
for _ in range(reps):
    ix.search(Q, 10)
pr = ix.get_search_profile(); ix.set_search_profile(False)
ref = Q.double() @ rows.double().T
rv, ri = torch.topk(ref, 10, dim=1)
differ = ids != ri
gap = float((rv - torch.gather(ref, 1, ids)).abs()[differ].max()) if bool(differ.any()) else 0.0
print(json.dumps({"queries": nq, "ms_per_search": round(e0.elapsed_time(e1) / reps, 4), "stages_ms": {k: round(v, 4) for k, v in pr.items() if k != "calls"},
                  "per_search": {k: v / reps for k, v in st.items()}, "ids_differ_only_within_fp64_gap": gap, "ok": gap < 3e-7}))
