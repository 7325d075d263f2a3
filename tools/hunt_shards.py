"""Bug hunt: N random shardings of a random index (1..8 shards of uneven sizes, duplicate rows placed across shard borders,
random rows / queries / width / k): per-shard vr_index_search_keys with the shard's id offset + vr_topk_merge_keys against ONE
index over all rows — scores and ids bit-identical.    python tools/hunt_shards.py 120        (round 3: 120 shardings, 0 failures)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from visrag_amd.engine import HipIndex, topk_merge_keys

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = []
for seed in range(n):
    rng = np.random.default_rng(seed)
    dim = int(rng.choice([64, 128, 256, 512, 2304])); nd = int(rng.choice([40, 333, 2000, 9000, 25000]))
    nq = int(rng.choice([1, 5, 16, 17, 100, 300])); k = int(rng.choice([1, 5, 10, 26, 40, 100])); P = int(rng.integers(1, 9))
    C = rng.standard_normal((nd, dim)).astype(np.float32); C /= np.linalg.norm(C, axis=1, keepdims=True)
    Q = rng.standard_normal((nq, dim)).astype(np.float32); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    cuts = sorted(set([0, nd] + rng.integers(1, nd, P - 1).tolist())) if P > 1 else [0, nd]
    for c in cuts[1:-1]:                                      # an exact duplicate pair straddling every border
        C[min(c + int(rng.integers(0, 3)), nd - 1)] = C[max(c - 1 - int(rng.integers(0, 3)), 0)]
    Q[0] = C[cuts[len(cuts) // 2] - 1]                        # and a query that hits such a pair
    q = torch.from_numpy(Q).cuda()
    try:
        full = HipIndex(dim, nd); full.add(C)
        fs, fi = full.search(q, k)
        parts = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            sh = HipIndex(dim, hi - lo); sh.add(C[lo:hi])
            parts.append(sh.search_keys(q, k, id_offset=lo))
        ms, mi = topk_merge_keys(torch.stack(parts))
        if not (torch.equal(mi, fi) and torch.equal(ms, fs)):
            bad.append((seed, dim, nd, nq, k, cuts, int((mi != fi).sum())))
    except Exception as ex:
        bad.append((seed, dim, nd, nq, k, cuts, repr(ex)[:200]))
print("cases", n, "failures", len(bad))
for b in bad[:10]: print(b)
