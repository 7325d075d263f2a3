import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visrag_amd.engine import HipIndex
from oracle import visrag_ret_oracle as O
def unit(n, d, seed):
    rng = np.random.default_rng(seed); x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)
for nd, nq, dim, k in [(3001, 257, 128, 26), (40037, 1000, 512, 10), (255, 129, 64, 3), (3001, 257, 128, 10), (3001, 140, 128, 26)]:
    C, Q = unit(nd, dim, 1), unit(nq, dim, 2)
    ix = HipIndex(dim, nd); ix.add(C)
    sc, ids = ix.search(Q, k)
    rs, ri = O.search_topk(Q, C, k)
    bad = np.argwhere(ids != ri)
    print(nd, nq, dim, k, "mismatches", len(bad), bad[:6].tolist())
    for q, c in bad[:3]:
        print("  q", q, "col", c, "got", ids[q, max(0,c-1):c+3].tolist(), sc[q, max(0,c-1):c+3].tolist(), "ref", ri[q, max(0,c-1):c+3].tolist(), rs[q, max(0,c-1):c+3].tolist())
