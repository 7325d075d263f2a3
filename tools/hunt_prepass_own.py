"""Bug hunt for the threshold pre-pass that owns its sample (search.hip: search_prepass_own_kernel / search_thr_own_kernel) and the
tile-skipping sweep (search256w.hip): random (rows, queries, width, k) among the shapes whose plan says prepass_chunks = 8, with
structure thrown at the sampled tiles — near-duplicate clusters inside / across / next to them, exact duplicates (ties at the
threshold: the own lists overflow), queries that are sampled rows — against an fp64 brute force.   python tools/hunt_prepass_own.py 60"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visrag_amd.engine import HipIndex
import tests.test_gpu_search as T

n, bad, done, seen = int(sys.argv[1]) if len(sys.argv) > 1 else 40, [], 0, {"flagged": 0, "certified": 0, "certified_extended": 0, "exact_pass": 0}
seed = 0
while done < n and seed < 20 * n:
    seed += 1
    rng = np.random.default_rng(11000 + seed)
    dim = int(rng.choice([128, 256, 384, 512]))
    nd = int(rng.integers(32600, 125000))
    nq = int(rng.choice([40, 130, 300, 520, 777, 1000, 1500]))
    k = int(rng.choice([1, 5, 10, 16, 26]))
    ix = HipIndex(dim, nd)
    C = T._unit(nd, dim, 12000 + seed)
    Q = T._unit(nq, dim, 13000 + seed)
    spot = T._prepass_spot_rows(nd, range(16))
    for _ in range(int(rng.integers(0, 5))):        # clusters: inside a sampled tile, straddling its edge, or anywhere
        n_dup = int(rng.integers(2, 400))
        where = rng.random()
        if where < 0.4:
            lo = int(rng.choice(spot[:-n_dup] if len(spot) > n_dup else spot[:1]))
        elif where < 0.7:
            edge = int(rng.choice([spot[0], spot[255] + 1, spot[256 * 5], spot[256 * 9 + 255] + 1]))
            lo = max(0, edge - n_dup // 2)
        else:
            lo = int(rng.integers(0, nd - n_dup))
        lo = min(lo, nd - n_dup)
        base = Q[int(rng.integers(nq))] + float(rng.uniform(0.2, 0.9)) * T._unit(1, dim, int(rng.integers(1 << 30)))[0]
        base /= np.linalg.norm(base)
        C[lo:lo + n_dup] = base[None, :] + float(rng.choice([0.0, 1e-5, 1e-4, 1e-3])) * rng.standard_normal((n_dup, dim)).astype(np.float32)
        C[lo:lo + n_dup] /= np.linalg.norm(C[lo:lo + n_dup], axis=1, keepdims=True)
    if rng.random() < 0.3:                          # > 1024 exact copies of a query's best row inside sampled tiles: the own lists overflow
        rows = T._prepass_spot_rows(nd, rng.choice(16, 5, replace=False).tolist())
        C[rows] = Q[int(rng.integers(nq))]
    for _ in range(int(rng.integers(0, 3))):
        Q[int(rng.integers(nq))] = C[int(rng.choice(spot))]
    ix.add(C)
    plan = ix.search_plan(nq)
    if plan["prepass_chunks"] != 8:
        ix.close(); continue
    done += 1
    ix.search_stats(reset=True)
    try:
        sc, ids = ix.search(Q, k)
        st = ix.search_stats()
        T._assert_ids_equal_fp64(ids, sc, C, Q, k)
        assert st["uncertified"] == 0 and st["certified"] + st["certified_extended"] + st["flagged"] == nq, st
        for key in seen: seen[key] += st[key]
    except Exception as e:
        bad.append((seed, nd, nq, dim, k, repr(e)[:300]))
    ix.close()
print("shapes:", done, "failures:", len(bad), seen)
for b in bad[:10]: print(b)
