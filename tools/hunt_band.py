"""Bug hunt for the flagged-query fallbacks (search_band.hip / search_exact.hip): random corpora with a CONTIGUOUS block of
near-duplicates (1 100 .. 9 500 rows at 1e-5 .. 1e-3 around a direction close to some queries), sometimes a second scattered
cluster and exact duplicates, random (rows, queries, width, k), against an fp64 brute force; every query must end certified or
redone (band pass / exact pass), ids equal up to fp32 summation noise.      python tools/hunt_band.py 40"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visrag_amd.engine import HipIndex
import tests.test_gpu_search as T

n, bad, seen = int(sys.argv[1]) if len(sys.argv) > 1 else 40, [], {"band_pass": 0, "exact_pass": 0, "certified": 0, "certified_extended": 0}
for seed in range(n):
    rng = np.random.default_rng(7000 + seed)
    dim = int(rng.choice([256, 512, 1152, 2304]))
    nd = int(rng.integers(9000, 40000))
    nq = int(rng.choice([1, 3, 16, 17, 40, 300]))
    k = int(rng.choice([1, 5, 10, 26, 27, 60]))
    C, Q = T._unit(nd, dim, 8000 + seed), T._unit(nq, dim, 9000 + seed)
    n_dup = int(rng.integers(1100, min(9500, nd - 100)))
    lo = int(rng.integers(0, nd - n_dup))
    base = Q[int(rng.integers(nq))] + float(rng.uniform(0.2, 0.8)) * T._unit(1, dim, seed)[0]
    base /= np.linalg.norm(base)
    C[lo:lo + n_dup] = base[None, :] + float(rng.choice([1e-5, 1e-4, 1e-3])) * rng.standard_normal((n_dup, dim)).astype(np.float32)
    C[lo:lo + n_dup] /= np.linalg.norm(C[lo:lo + n_dup], axis=1, keepdims=True)
    if rng.random() < 0.5:
        rows = rng.choice(nd, 60, replace=False)
        C[rows] = C[rows[0]]                                   # exact duplicates, scattered
    try:
        ix = HipIndex(dim, nd); ix.add(C)
        ix.search_stats(reset=True)
        sc, ids = ix.search(Q, k)
        st = ix.search_stats()
        T._assert_ids_equal_fp64(ids, sc, C, Q, k)
        assert st["uncertified"] == 0 and st["certified"] + st["certified_extended"] + st["flagged"] == nq, st
        for key in seen: seen[key] += st[key]
        ix.close()
    except Exception as e:
        bad.append((seed, dim, nd, nq, k, n_dup, repr(e)[:300]))
print("configs", n, "failures", len(bad), "outcomes", seen)
for b in bad[:10]: print(b)
