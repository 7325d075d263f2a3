"""-m gpu: fused similarity + top-k (vr_index_*) against the oracle's fp32 matmul + top-k
(oracle/visrag_ret_oracle.py::search_topk == dense_retriever.py:13-34 with a fixed tie rule).
Bar: identical ids (bit-exact index work), scores within 1e-5 (fp32 dot, different summation
order)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import visrag_ret_oracle as O  # noqa: E402
from visrag_amd.engine import HipIndex, topk_merge  # noqa: E402
from visrag_amd.retriever import merge_topk_host  # noqa: E402


def _unit(n, d, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.mark.parametrize("nd,nq,dim,k", [(5000, 37, 256, 10), (130, 3, 64, 5), (20000, 300, 2304, 10),
                                          (1000, 130, 128, 20), (7, 2, 64, 10),
                                          (3001, 257, 128, 26), (40037, 1000, 512, 10), (255, 129, 64, 3),
                                          (20000, 16, 512, 10), (3000, 1, 256, 26), (7, 2, 256, 10), (50001, 9, 2304, 10)])
def test_search_matches_oracle(nd, nq, dim, k):
    C, Q = _unit(nd, dim, 1), _unit(nq, dim, 2)
    ix = HipIndex(dim, nd)
    ix.add(C[: nd // 2]); ix.add(C[nd // 2:])
    assert len(ix) == nd
    sc, ids = ix.search(Q, k)
    rs, ri = O.search_topk(Q, C, k)
    kk = min(k, nd)
    np.testing.assert_allclose(sc[:, :kk], rs, atol=1e-5, rtol=0)
    # ids identical, except where two fp32 scores are within summation-order rounding of each other
    for q, c in np.argwhere(ids[:, :kk] != ri):
        j = int(np.flatnonzero(ri[q] == ids[q, c])[0]) if ids[q, c] in ri[q] else -1
        assert j >= 0 and abs(rs[q, j] - rs[q, c]) < 3e-7, (q, c, ids[q, c], ri[q, c])
    if kk < k:
        assert (ids[:, kk:] == -1).all() and np.isinf(sc[:, kk:]).all()


def test_search_ties_and_duplicates():
    """Duplicate rows tie exactly: lower row id must win, like the oracle's rule."""
    dim = 128
    C = _unit(2000, dim, 3)
    C[1500:1540] = C[7]                       # 40 copies of row 7 -> more ties than candidates kept
    C[900] = C[11]
    Q = np.concatenate([C[7:8], C[11:12], _unit(5, dim, 4)])
    ix = HipIndex(dim, 2000)
    ix.add(C)
    sc, ids = ix.search(Q, 10)
    rs, ri = O.search_topk(Q, C, 10)
    assert np.array_equal(ids, ri)
    assert ids[0, 0] == 7 and list(ids[0, 1:]) == list(range(1500, 1509))
    assert list(ids[1, :2]) == [11, 900]


@pytest.mark.parametrize("nq,dim", [(7, 64), (7, 256), (200, 64)])
def test_search_adversarial_orders(nq, dim):
    """Corpora that defeat the threshold pre-pass: (a) every row identical (all scores tie: the
    lowest ids must win, every candidate list overflows), (b) scores increasing with the row id
    (each new row beats everything before it).  nq=200 runs the 256-tile sweep (wave-owned
    half-lists + own compaction + the merge fallback), nq=7 the 128-tile sweep (dim 64) or the
    streaming kernel (dim 256)."""
    nd, k = 20000, 10
    rng = np.random.default_rng(5)
    v = _unit(1, dim, 9)[0]
    Q = v[None, :] + 0.02 * rng.standard_normal((nq, dim)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    # (a) identical rows
    C = np.tile(_unit(1, dim, 6), (nd, 1))
    ix = HipIndex(dim, nd); ix.add(C)
    sc, ids = ix.search(Q, k)
    assert np.array_equal(ids, np.tile(np.arange(k), (nq, 1)))
    np.testing.assert_allclose(sc, (Q @ C[0])[:, None].repeat(k, 1), atol=1e-5)
    # (b) random unit rows sorted by their score against the queries' common direction: for every
    #     query the scores rise (almost) monotonically with the row id
    C = _unit(nd, dim, 8)
    C = C[np.argsort(C @ v, kind="stable")]
    ix = HipIndex(dim, nd); ix.add(C)
    sc, ids = ix.search(Q, k)
    rs, ri = O.search_topk(Q, C, k)
    np.testing.assert_allclose(sc, rs, atol=1e-5, rtol=0)
    for q, c in np.argwhere(ids != ri):
        j = int(np.flatnonzero(ri[q] == ids[q, c])[0]) if ids[q, c] in ri[q] else -1
        assert j >= 0 and abs(rs[q, j] - rs[q, c]) < 3e-7, (q, c)


def test_search_golden_reference(golden_dir):
    """Same corpus/queries as the REFERENCE run in tests/golden/retrieve.npz: the global top-5
    must be the reference's (distributed_parallel_retrieve over 3 shards)."""
    g = np.load(os.path.join(golden_dir, "retrieve.npz"))
    ix = HipIndex(64, 600)
    ix.add(g["C"])
    sc, ids = ix.search(g["Q"], 5)
    for qi in range(len(g["Q"])):
        assert [f"doc{j}" for j in ids[qi]] == [str(d) for d in g["docs"][qi][:5]]
        np.testing.assert_allclose(sc[qi], g["scores"][qi][:5], atol=1e-5)


def test_search_device_tensors_and_sortedness():
    """Full BASELINE size (100k x 2304, 1k queries): size-independent properties — scores
    sorted, ids valid and unique, every score equals the exact fp32 dot of its row, and
    agreement with a torch fp32 brute force on a query subset."""
    nd, nq, dim, k = 100_000, 1000, 2304, 10
    g = torch.Generator(device="cuda").manual_seed(0)
    C = torch.randn((nd, dim), generator=g, device="cuda")
    C = C / C.norm(dim=1, keepdim=True)
    Q = torch.randn((nq, dim), generator=g, device="cuda")
    Q = Q / Q.norm(dim=1, keepdim=True)
    ix = HipIndex(dim, nd)
    ix.add(C)
    sc, ids = ix.search(Q, k)
    torch.cuda.synchronize()
    assert (sc[:, :-1] >= sc[:, 1:]).all()
    assert (ids >= 0).all() and (ids < nd).all()
    assert all(len(set(r.tolist())) == k for r in ids[:50].cpu())
    exact = (Q[:, None, :] * C[ids]).sum(-1)
    assert (exact - sc).abs().max() < 2e-6
    # EVERY one of the 1000 queries against an fp64 brute force over all 100k rows (k = 10 and the deep
    # path k = 100): ids identical, except where two fp64 scores differ by less than fp32 summation noise
    ref = Q.double() @ C.double().T
    for kk in (10, 100):
        s2, i2 = (sc, ids) if kk == k else ix.search(Q, kk)
        rv, ri = torch.topk(ref, kk, dim=1)
        assert (s2.double() - torch.gather(ref, 1, i2)).abs().max() < 2e-7          # returned scores are exact dots
        bad = i2 != ri
        if bool(bad.any()):
            gap = (rv - torch.gather(ref, 1, i2)).abs()[bad].max()
            assert float(gap) < 1e-7, (kk, int(bad.any(dim=1).sum()), float(gap))
            assert int(bad.any(dim=1).sum()) <= 10


@pytest.mark.parametrize("nd,nq,dim,k", [(5000, 37, 256, 27), (20000, 300, 512, 100), (700, 5, 64, 1000),
                                          (100000, 64, 2304, 100), (3000, 260, 128, 64)])
def test_deep_retrieval_matches_oracle(nd, nq, dim, k):
    """k > 26 (TREC-depth retrieval): GEMM + radix select + exact re-score path."""
    C, Q = _unit(nd, dim, 11), _unit(nq, dim, 12)
    ix = HipIndex(dim, nd)
    ix.add(C)
    sc, ids = ix.search(Q, k)
    rs, ri = O.search_topk(Q, C, k)
    kk = min(k, nd)
    np.testing.assert_allclose(sc[:, :kk], rs, atol=1e-5, rtol=0)
    for q, c in np.argwhere(ids[:, :kk] != ri):
        j = int(np.flatnonzero(ri[q] == ids[q, c])[0]) if ids[q, c] in ri[q] else -1
        assert j >= 0 and abs(rs[q, j] - rs[q, c]) < 3e-7, (q, c, ids[q, c], ri[q, c])
    if kk < k:
        assert (ids[:, kk:] == -1).all() and np.isinf(sc[:, kk:]).all()


def test_deep_retrieval_ties():
    """All rows identical: the k lowest ids, in order (ties at the selection threshold)."""
    dim, nd, k = 64, 5000, 200
    C = np.tile(_unit(1, dim, 6), (nd, 1))
    Q = _unit(3, dim, 7)
    ix = HipIndex(dim, nd); ix.add(C)
    sc, ids = ix.search(Q, k)
    assert np.array_equal(ids, np.tile(np.arange(k), (3, 1)))
    with pytest.raises(Exception):
        ix.search(Q, 1001)


@pytest.mark.parametrize("P,nq,k", [(4, 33, 10), (8, 17, 100), (3, 5, 1000)])
def test_topk_merge_matches_host_rule(P, nq, k):
    rng = np.random.default_rng(5)
    sc = np.sort(rng.standard_normal((P, nq, k)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    ids = rng.permutation(P * nq * k).reshape(P, nq, k).astype(np.int64)
    sc[1, :, 3] = sc[0, :, 2]                 # cross-part ties
    ids[2, nq // 2, 7:] = -1; sc[2, nq // 2, 7:] = -np.inf
    ms, mi = topk_merge(torch.from_numpy(sc).cuda(), torch.from_numpy(ids).cuda())
    hs, hi = merge_topk_host(sc, ids, k)
    assert np.array_equal(mi.cpu().numpy(), hi)
    np.testing.assert_array_equal(ms.cpu().numpy(), hs)


def test_search_on_a_side_stream():
    """First add/search of a fresh index issued on a non-default (non-blocking) torch stream: the
    library's buffers must be ready on THAT stream (allocation-time memsets run on the NULL stream)."""
    C, Q = _unit(30000, 256, 21), _unit(300, 256, 22)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ix = HipIndex(256, 30000)
        ix.add(torch.from_numpy(C).cuda())
        sc, ids = ix.search(torch.from_numpy(Q).cuda(), 10)
        sc1, ids1 = ix.search(torch.from_numpy(Q[:3]).cuda(), 10)
    s.synchronize()
    rs, ri = O.search_topk(Q, C, 10)
    assert np.array_equal(ids.cpu().numpy(), ri) and np.array_equal(ids1.cpu().numpy(), ri[:3])
    np.testing.assert_allclose(sc.cpu().numpy(), rs, atol=1e-5, rtol=0)


SHARD_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["VR_ROOT"])
import numpy as np, torch, torch.distributed as dist
from visrag_amd.engine import HipIndex
from visrag_amd.retriever import sharded_search
rank, world = int(os.environ["VR_RANK"]), 2
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["VR_PORT"], rank=rank, world_size=world)
torch.cuda.set_device(0)
rng = np.random.default_rng(0)
nd, nq, dim, k = 30011, 333, 512, 10
C = rng.standard_normal((nd, dim)).astype(np.float32); C /= np.linalg.norm(C, axis=1, keepdims=True)
Q = rng.standard_normal((nq, dim)).astype(np.float32); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
C[20000] = C[5]                                          # exact tie across the two shards
per = (nd + world - 1) // world
lo, hi = rank * per, min(nd, (rank + 1) * per)
shard = HipIndex(dim, hi - lo); shard.add(torch.from_numpy(C[lo:hi]).cuda())     # this rank's rows only
q = torch.from_numpy(Q).cuda()
sc, ids = sharded_search(shard, q, k, id_offset=lo)       # HipIndex.search + packed all-gather + vr_topk_merge
full = HipIndex(dim, nd); full.add(torch.from_numpy(C).cuda())
fs, fi = full.search(q, k)
torch.cuda.synchronize()
assert torch.equal(ids, fi), (rank, int((ids != fi).sum()))
assert torch.equal(sc, fs), rank
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sharded_search_two_ranks_on_one_gpu(tmp_path):
    """The product's multi-rank retrieval with world_size 2: two processes (sharing the one GPU of the
    test box, gloo rendezvous), each with a real HipIndex over ITS rows, retriever.sharded_search (local
    fused search -> one packed all-gather -> vr_topk_merge) == a single index over all rows, bit for bit."""
    import socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(SHARD_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, VR_ROOT=root, VR_PORT=str(port), VR_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_retrieve_trec_equals_reference(golden_dir, tmp_path):
    """The unchanged driver's retrieve phase (driver/eval.py:215-232): pickle shards -> retriever.
    distributed_parallel_retrieve(args, k) with DEFAULT arguments -> save_as_trec.  Against the TREC file the
    REFERENCE wrote for the same shards (tests/golden/retrieve.npz: 3 corpus shards, k = 5, i.e. 15 docs per
    query — the union of the per-shard top-k, dense_retriever.py:79-92): same lines in the same order,
    scores equal to 1e-5 (fp32 dots in a different summation order)."""
    import types
    from visrag_amd import utils as U
    from visrag_amd.retriever import distributed_parallel_retrieve
    g = np.load(os.path.join(golden_dir, "retrieve.npz"))
    C, Q = g["C"], g["Q"]
    n, per = len(C), (len(C) + 2) // 3
    for s in range(3):
        lo, hi = s * per, min(n, (s + 1) * per)
        U.write_shard(os.path.join(tmp_path, U.shard_name("corpus", 0, lo, hi)), C[lo:hi], [f"doc{j}" for j in range(lo, hi)])
    U.write_shard(os.path.join(tmp_path, U.shard_name("query", 0)), Q, [f"q{j}" for j in range(len(Q))])
    args = types.SimpleNamespace(output_dir=str(tmp_path), process_index=0, device="cuda:0")
    run = distributed_parallel_retrieve(args, 5)
    out = os.path.join(tmp_path, "trec", "test.0.trec")
    U.save_as_trec(run, out)
    got = [l.split("\t") for l in open(out).read().strip().split("\n")]
    ref = [l.split("\t") for l in str(g["trec"]).strip().split("\n")]
    assert len(got) == len(ref) == len(Q) * 15
    for a, b in zip(got, ref):
        assert a[:4] == b[:4] and a[5] == b[5], (a, b)              # qid, Q0, docid, rank, run id
        assert abs(float(a[4]) - float(b[4])) < 1e-5, (a, b)
    top = distributed_parallel_retrieve(args, 5, global_topk=True)
    assert all(len(v) == 5 and set(v) <= set(run[q]) for q, v in top.items())


def test_retrieve_phase_equals_the_reference_drivers_output(golden_dir, tmp_path):
    """Row f2 on the GPU: the statements of driver/eval.py's retrieve phase (:210-304 — distributed_parallel_retrieve,
    save_as_trec, glob + load_from_trec, pytrec_eval ndcg_cut.10 / recall.10, eval_mrr(...)['all']) with the swapped
    imports of INTEGRATION.md, over the real HipIndex.  Against what the REFERENCE's own eval.py wrote and printed for the
    same shards and qrels (tests/golden/eval_driver.npz, oracle/gen_golden.py --evaldriver): the TREC lines (scores to 1e-5:
    fp32 dots in a different summation order), test_result.log and the three metric lines, character for character."""
    import glob, types
    from test_cpu_eval_dropin import _write_case
    from visrag_amd import pytrec_eval
    from visrag_amd.retriever import distributed_parallel_retrieve
    from visrag_amd.utils import eval_mrr, load_from_trec, save_as_trec
    g = np.load(os.path.join(golden_dir, "eval_driver.npz"))
    out = str(tmp_path)
    qrels_path = _write_case(out, golden_dir)
    assert open(qrels_path).read() == str(g["qrels"])
    qrels = {}
    for ln in open(qrels_path).read().strip().split("\n")[1:]:                       # load_beir_qrels, eval.py:58-68
        q, d, r = ln.split("\t")
        qrels.setdefault(q, {})[d] = int(r)
    args = types.SimpleNamespace(output_dir=out, process_index=0, world_size=1, retrieve_depth=10, device="cuda:0")
    run = distributed_parallel_retrieve(args=args, topk=args.retrieve_depth)
    save_as_trec(run, os.path.join(out, "test.0.trec"))
    got = [l.split("\t") for l in open(os.path.join(out, "test.0.trec")).read().strip().split("\n")]
    ref = [l.split("\t") for l in str(g["trec"]).strip().split("\n")]
    assert len(got) == len(ref) > 0
    for a, b in zip(got, ref):
        assert a[:4] == b[:4] and a[5] == b[5] and abs(float(a[4]) - float(b[4])) < 1e-5, (a, b)
    run = {}
    for part in glob.glob(os.path.join(out, "test.*.trec")):
        run.update(load_from_trec(part))
    ev = pytrec_eval.RelevanceEvaluator(qrels, {"ndcg_cut.10", "recall.10"}).evaluate(run)
    lines = []
    for measure in sorted(sorted(ev.items())[-1][1].keys()):
        lines.append("{:25s}{:8s}{:.4f}".format(measure, "all", pytrec_eval.compute_aggregated_measure(measure, [m[measure] for m in ev.values()])))
    lines.append(f"MRR@10: {eval_mrr(qrels, run, 10)['all']}")
    assert lines == [str(x) for x in g["lines"]], (lines, g["lines"])
    assert lines[1] + "\n" == str(g["log"])


def _assert_ids_equal_fp64(ids, sc, C, Q, k):
    """ids == the fp64 brute force's, except where two fp64 scores are closer than fp32 summation noise."""
    ref = Q.astype(np.float64) @ C.astype(np.float64).T
    order = np.lexsort((np.broadcast_to(np.arange(ref.shape[1]), ref.shape), -ref), axis=1)[:, :k]
    rs = np.take_along_axis(ref, order, 1)
    np.testing.assert_allclose(sc, np.take_along_axis(ref, ids, 1), atol=2e-6, rtol=0)       # scores are the rows' exact dots
    for q, c in np.argwhere(ids != order):
        assert abs(ref[q, ids[q, c]] - rs[q, c]) < 3e-7, (q, c, ids[q, c], order[q, c], ref[q, ids[q, c]], rs[q, c])
    assert (np.diff(sc, axis=1) <= 0).all()


@pytest.mark.parametrize("n_dup,nq,k", [(20, 1, 10), (64, 1, 10), (300, 1, 10), (20, 40, 10), (64, 40, 10), (300, 300, 10),
                                        (300, 5, 26), (200, 3, 60), (2000, 40, 100), (2000, 1, 10), (2000, 40, 10)])
def test_search_near_duplicate_cluster_is_exact(n_dup, nq, k):
    """north_star: IDENTICAL top-k doc ids.  A cluster of near-duplicate pages (within 1e-4 of each other: less
    than the bf16 dot-product error, so the bf16 sweep orders them at random) sits at the top of query 0's
    ranking, with more members than the sweep keeps candidates (16 / 32 / k + 24).  The certification
    (search_common.h) must notice that its candidates cannot prove the fp32 top-k and either re-score further
    candidates (n_dup = 20) or send the query through the exact fp32 pass (search_exact.hip): ids equal to an fp64
    brute force, for the streaming kernel (nq <= 16), the 256-tile sweep and the deep path (k > 26)."""
    dim, nd = 2304, 20000
    C = _unit(nd, dim, 31)
    Q = _unit(nq, dim, 32)
    rng = np.random.default_rng(33)
    base = Q[0] + 0.5 * _unit(1, dim, 34)[0]
    base /= np.linalg.norm(base)
    cluster = 5000 + 7 * np.arange(n_dup)                                   # spread over many chunks
    C[cluster] = base[None, :] + 1e-4 * rng.standard_normal((n_dup, dim)).astype(np.float32)
    C[cluster] /= np.linalg.norm(C[cluster], axis=1, keepdims=True)
    ix = HipIndex(dim, nd); ix.add(C)
    ix.search_stats(reset=True)
    sc, ids = ix.search(Q, k)
    st = ix.search_stats()
    assert set(ids[0, :min(k, n_dup)].tolist()) <= set(cluster.tolist())
    _assert_ids_equal_fp64(ids, sc, C, Q, k)
    assert st["certified"] + st["certified_extended"] + st["flagged"] == nq and st["uncertified"] == 0, st
    if n_dup == 20 and k == 10:
        assert st["certified_extended"] >= 1 and st["flagged"] == 0, st      # 4 more candidates re-scored, nothing behind the sweep
    if n_dup >= 2000:                                                        # beyond what the merge re-scores in place (1024 rows):
        assert st["band_pass"] >= 1 and st["exact_pass"] == 0, st            # the band pass re-scores the 2000, no sweep of the fp32 index
    elif k <= 26 and nq <= 16:
        # (with more than 16 queries the sweep starts from pre-pass thresholds; a cluster this dense pulls query 0's
        # threshold up into its own error band, the lists then cannot prove completeness and the band pass is right)
        assert st["flagged"] == 0, st
    # the same search with certification off is what rounds 1-2 shipped: tolerance-exact only
    ix.set_search_eps(-1.0)
    sc2, ids2 = ix.search(Q, k)
    assert ix.search_stats()["uncertified"] == nq
    exact = C.astype(np.float64) @ Q[0].astype(np.float64)
    assert (exact[ids2[0]] >= np.sort(exact)[-k] - 1e-3).all()


@pytest.mark.parametrize("nd,nq,dim,k", [(20000, 1, 512, 10), (20000, 16, 2304, 10), (20000, 40, 256, 10), (30000, 300, 512, 26),
                                          (5000, 37, 256, 40), (100, 3, 64, 10), (7, 2, 64, 10)])
def test_exact_pass_alone_matches_oracle(nd, nq, dim, k):
    """An error model so pessimistic that NO query can be certified: every query is flagged, its error band is the whole
    index, and the result comes from the band pass alone (indices of up to 8192 rows: every row re-scored by
    band_select_kernel), from the exact fp32 pass alone (larger ones: exact_scores_kernel + radix select) or — a handful of
    queries on the streaming search — from the merge workgroup's own walk over its score row in segments of 8192 rows."""
    C, Q = _unit(nd, dim, 41), _unit(nq, dim, 42)
    ix = HipIndex(dim, nd); ix.add(C)
    ix.set_search_eps(100.0)
    ix.search_stats(reset=True)
    sc, ids = ix.search(Q, k)
    st = ix.search_stats()
    kk = min(k, nd)
    if nd > k + 24:                                              # (bands beyond 8192 rows: counted as exact_pass whoever walks them —
        assert st["flagged"] == nq and (st["exact_pass"] == nq if nd > 8192 else st["band_pass"] == nq), st   # the first ones of an index: one workgroup each)
    _assert_ids_equal_fp64(ids[:, :kk], sc[:, :kk], C, Q, kk)
    if kk < k:
        assert (ids[:, kk:] == -1).all() and np.isinf(sc[:, kk:]).all()
    # and the same queries under the default bound agree with it (random unit rows: certified without a full pass)
    ix.set_search_eps(None)
    ix.search_stats(reset=True)
    sc2, ids2 = ix.search(Q, k)
    assert np.array_equal(ids2, ids) and np.array_equal(sc2, sc)             # the same floats: ONE definition of the fp32 dot
    assert ix.search_stats()["uncertified"] == 0


def test_flagged_queries_walk_several_windows_of_the_score_buffer():
    """The fallback passes hold at most 512 MiB of fp32 score rows: `slots` = 2^27 / rows flagged queries per window, walked in
    a loop of launches (engine.hip: vr_index_search).  300 000 rows x 500 flagged queries = two windows for the band GEMM +
    band selection and for the exact pass (every band is the whole index: band pass hands on to the exact pass): ids ==
    fp64 for every query, whichever window it fell into."""
    nd, nq, dim, k = 300000, 500, 64, 10
    C, Q = _unit(nd, dim, 45), _unit(nq, dim, 46)
    ix = HipIndex(dim, nd); ix.add(C)
    ix.set_search_eps(100.0)
    ix.search_stats(reset=True)
    sc, ids = ix.search(Q, k)
    st = ix.search_stats()
    assert st["flagged"] == nq and st["exact_pass"] == nq and st["uncertified"] == 0, st
    _assert_ids_equal_fp64(ids, sc, C, Q, k)
    ix.close()


def test_random_index_is_certified_without_the_exact_pass():
    """BASELINE config 3's index (random unit rows, 100k x 2304, 1k queries, top-10) under the rigorous bound:
    every query certified from its candidate lists — the exact pass stays idle (its cost is what the
    flagged-query rate in the bench line watches)."""
    nd, nq, dim, k = 100_000, 1000, 2304, 10
    g = torch.Generator(device="cuda").manual_seed(3)
    C = torch.randn((nd, dim), generator=g, device="cuda"); C = C / C.norm(dim=1, keepdim=True)
    Q = torch.randn((nq, dim), generator=g, device="cuda"); Q = Q / Q.norm(dim=1, keepdim=True)
    ix = HipIndex(dim, nd); ix.add(C)
    ix.search_stats(reset=True)
    sc, ids = ix.search(Q, k)
    st = ix.search_stats()
    assert st["certified"] + st["certified_extended"] + st["flagged"] == nq
    assert st["flagged"] <= 5 and st["exact_pass"] == 0, st
    rv, ri = torch.topk(Q.double() @ C.double().T, k, dim=1)
    bad = ids != ri
    assert int(bad.any(dim=1).sum()) <= 3 and (not bool(bad.any()) or float((rv - torch.gather(Q.double() @ C.double().T, 1, ids)).abs()[bad].max()) < 1e-7)


@pytest.mark.parametrize("nd,nq,dim,k", [(2000, 1000, 2304, 10), (600, 300, 512, 10), (5000, 257, 1024, 10), (3000, 64, 2304, 16),
                                          (1500, 400, 256, 1), (7000, 1000, 768, 26), (2000, 130, 2304, 10), (900, 1000, 128, 10)])
def test_small_shards_are_searched_from_compacted_lists(nd, nq, dim, k):
    """A shard too small for the pre-pass threshold (fewer than 64 row tiles: the per-file searches of
    distributed_parallel_retrieve over small max_inmem_docs shards, an 8-way shard of a small corpus): every row enters a
    half-list and every half-list is compacted.  What compaction dropped is bounded per list (search256.hip: the largest
    KP-th best of a compacted half-list), not by the global KP-th best — which lies inside the error band of the k-th
    and sent four queries in five through the band pass until round 4.  ids == fp64 either way; the first case also
    watches the flagged share."""
    C, Q = _unit(nd, dim, 71), _unit(nq, dim, 72)
    ix = HipIndex(dim, nd); ix.add(C)
    ix.search_stats(reset=True)
    sc, ids = ix.search(Q, k)
    st = ix.search_stats()
    _assert_ids_equal_fp64(ids, sc, C, Q, k)
    assert st["uncertified"] == 0, st
    if (nd, nq, dim) == (2000, 1000, 2304):
        assert st["flagged"] < nq // 2, st


@pytest.mark.parametrize("nd", [1000, 12500, 20480, 32768, 50000, 70001])
@pytest.mark.parametrize("nq,dim,k", [(1, 2304, 10), (16, 2304, 10), (5, 512, 26), (9, 2560, 10)])
def test_streaming_kernel_strip_layouts(nd, nq, dim, k):
    """The handful-of-queries kernel (search_small.hip) gives each of its 256 workgroups ceil(nd / 256) rows (rounded to 16-row
    strips) and its 8 waves take one strip each per round.  Shard sizes with 1 / 4 / 5 / 8 / 13 / 18 strips per workgroup
    (full rounds, partial rounds, workgroups without rows, a ragged last strip) at three widths: ids == fp64, and the same
    on every run."""
    C, Q = _unit(nd, dim, 91), _unit(nq, dim, 92)
    ix = HipIndex(dim, nd); ix.add(C)
    ix.search_stats(reset=True)
    sc, ids = ix.search(Q, k)
    st = ix.search_stats()
    _assert_ids_equal_fp64(ids, sc, C, Q, k)
    assert st["uncertified"] == 0, st
    sc2, ids2 = ix.search(Q, k)
    assert np.array_equal(ids, ids2) and np.array_equal(sc, sc2)
    ix.close()


@pytest.mark.parametrize("nd,nq,dim,k", [(20000, 300, 512, 10), (3000, 7, 256, 26), (20000, 64, 512, 100), (5, 3, 64, 10)])
def test_search_keys_and_merge_keys(nd, nq, dim, k):
    """The packed exchange format (vr_index_search_keys / vr_topk_merge_keys): one 64-bit word per result, the
    shard's id offset applied by the search's own emit; merging the keys of two shards == one index."""
    from visrag_amd.engine import topk_merge_keys
    from visrag_amd.retriever import unpack_keys_host
    C, Q = _unit(nd, dim, 51), _unit(nq, dim, 52)
    if nd > 100:
        C[nd // 2 + 3] = C[5]                                              # exact tie across the two shards
    q = torch.from_numpy(Q).cuda()
    full = HipIndex(dim, nd); full.add(C)
    fs, fi = full.search(q, k)
    keys = full.search_keys(q, k, id_offset=1000)
    us, ui = unpack_keys_host(keys.cpu().numpy())
    assert np.array_equal(us, fs.cpu().numpy()) and np.array_equal(ui, np.where(fi.cpu().numpy() >= 0, fi.cpu().numpy() + 1000, -1))
    half = nd // 2
    parts = []
    for lo, hi in ((0, half), (half, nd)):
        sh = HipIndex(dim, max(hi - lo, 1)); sh.add(C[lo:hi])
        parts.append(sh.search_keys(q, k, id_offset=lo))
    ms, mi = topk_merge_keys(torch.stack(parts))
    assert torch.equal(mi, fi) and torch.equal(ms, fs)
    with pytest.raises(Exception):
        full.search_keys(q, k, id_offset=2 ** 32)


NCCL_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["VR_ROOT"])
import numpy as np, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", device_id=torch.device(f"cuda:{rank}"))        # RCCL
from visrag_amd.engine import HipIndex
from visrag_amd.retriever import sharded_search
N_GATHER = [0]
_ag = dist.all_gather_into_tensor
def _counted(out, inp, *a, **kw):
    assert inp.is_cuda and out.is_cuda and inp.dtype == torch.int64          # device key buffers over RCCL
    N_GATHER[0] += 1
    return _ag(out, inp, *a, **kw)
dist.all_gather_into_tensor = _counted
rng = np.random.default_rng(0)
nd, nq, dim, k = 30011, 333, 512, 10
C = rng.standard_normal((nd, dim)).astype(np.float32); C /= np.linalg.norm(C, axis=1, keepdims=True)
Q = rng.standard_normal((nq, dim)).astype(np.float32); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
C[20000] = C[5]
per = (nd + world - 1) // world
lo, hi = rank * per, min(nd, (rank + 1) * per)
shard = HipIndex(dim, hi - lo, device=rank); shard.add(torch.from_numpy(C[lo:hi]).cuda())
q = torch.from_numpy(Q).cuda()
sc, ids = sharded_search(shard, q, k, id_offset=lo)       # search_keys -> ONE RCCL all-gather of device keys -> vr_topk_merge_keys
assert dist.get_backend() == "nccl" and sc.is_cuda
full = HipIndex(dim, nd, device=rank); full.add(torch.from_numpy(C).cuda())
fs, fi = full.search(q, k)
torch.cuda.synchronize()
assert torch.equal(ids, fi) and torch.equal(sc, fs), rank
# the same transport behind the kept API: distributed_parallel_retrieve, corpus-sharded == replicated
import types
from visrag_amd.retriever import distributed_parallel_retrieve
from visrag_amd.utils import shard_name, write_shard
out = os.environ["VR_OUT"]
write_shard(os.path.join(out, shard_name("corpus", rank)), C[lo:hi], [f"d{i}" for i in range(lo, hi)])
qper = (nq + world - 1) // world
write_shard(os.path.join(out, shard_name("query", rank)), Q[rank * qper:(rank + 1) * qper], [f"q{i}" for i in range(rank * qper, min(nq, (rank + 1) * qper))])
dist.barrier()
args = types.SimpleNamespace(output_dir=out, process_index=rank, device=f"cuda:{rank}")
for gt in (False, True):
    ref = distributed_parallel_retrieve(args, k, global_topk=gt, sharded=False)
    got = distributed_parallel_retrieve(args, k, global_topk=gt, sharded=True)
    assert got == ref and list(got) == list(ref), (rank, gt)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok", "all_gathers", N_GATHER[0])
'''


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_sharded_search_over_rccl(tmp_path):
    """The multi-GPU retrieval path on its real transport: one process per GPU, backend "nccl" (= RCCL over xGMI),
    the all-gather on DEVICE key buffers.  Skipped on the 1-GPU boxes of this pool; there the same function runs
    over gloo (test_sharded_search_two_ranks_on_one_gpu)."""
    import socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(NCCL_WORKER)
    n = min(torch.cuda.device_count(), 4)
    procs = []
    for r in range(n):
        env = dict(os.environ, VR_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_RANK=str(r), HSA_ENABLE_IPC_MODE_LEGACY="0", VR_OUT=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_sharded_search_over_rccl_world_of_one(tmp_path):
    """The RCCL code path on ONE GPU: backend "nccl" with a single rank runs the same statements as N ranks —
    `HipIndex.search_keys` -> ONE `all_gather_into_tensor` of device int64 keys (retriever.exchange_keys) -> `vr_topk_merge_keys`
    — for `sharded_search` and for the corpus-sharded `distributed_parallel_retrieve` (replicated and global top-k): three
    collectives on device buffers, results equal to the unsharded search.  (Two ranks need two GPUs: the test above.)"""
    import socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(NCCL_WORKER)
    env = dict(os.environ, VR_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", VR_OUT=str(tmp_path), VISRAG_SHARDED_RETRIEVE="force")   # a world of one skips the transport unless forced
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout
    assert "rank 0 ok all_gathers 3" in p.stdout, p.stdout          # sharded_search + the two corpus-sharded retrieves


@pytest.mark.parametrize("nq", [1, 40])
def test_wide_band_is_rescored_inside_the_merge(nq):
    """100 near-duplicates at the top (more than the 64 sorted candidates, fewer than the merge's 1024-entry buffer): every
    row inside the error band is re-scored in the merge kernel — exact ids without the pass over the whole index; four
    exact copies of a page tie and come out lowest id first."""
    dim, nd, k, n_dup = 2304, 20000, 10, 100
    C = _unit(nd, dim, 61)
    Q = _unit(nq, dim, 62)
    rng = np.random.default_rng(63)
    base = Q[0] + 0.5 * _unit(1, dim, 64)[0]
    base /= np.linalg.norm(base)
    cluster = 3000 + 11 * np.arange(n_dup)
    C[cluster] = base[None, :] + 1e-4 * rng.standard_normal((n_dup, dim)).astype(np.float32)
    C[cluster] /= np.linalg.norm(C[cluster], axis=1, keepdims=True)
    C[[17000, 17001, 17002]] = C[cluster[0]]                                  # exact copies
    ix = HipIndex(dim, nd); ix.add(C)
    ix.search_stats(reset=True)
    sc, ids = ix.search(Q, k)
    st = ix.search_stats()
    _assert_ids_equal_fp64(ids, sc, C, Q, k)
    assert st["flagged"] == 0 and st["regathered"] >= 1, st


@pytest.mark.parametrize("seed", list(range(24)))
def test_search_random_shapes_and_structure_vs_fp64(seed):
    """Seeded random (rows, queries, width, k) with random structure thrown in — exact duplicate rows, near-duplicate
    clusters at 1e-5 .. 1e-3, queries that ARE index rows, two index parts — against an fp64 brute force: identical
    ids up to fp32 summation noise, every query certified or sent through the exact pass."""
    rng = np.random.default_rng(1000 + seed)
    dim = int(rng.choice([64, 128, 192, 256, 384, 768, 1152, 2304]))
    nd = int(rng.choice([1, 9, 130, 257, 1000, 4097, 12000, 30011]))
    nq = int(rng.choice([1, 2, 7, 16, 17, 33, 128, 129, 300]))
    k = int(rng.choice([1, 3, 10, 16, 26, 27, 60, 100]))
    C = _unit(nd, dim, 2000 + seed)
    Q = _unit(nq, dim, 3000 + seed)
    if nd >= 130:
        for _ in range(int(rng.integers(0, 4))):                     # clusters around a direction close to some query
            n_dup = int(rng.integers(2, min(80, nd // 2)))
            base = Q[int(rng.integers(nq))] + float(rng.uniform(0.2, 1.0)) * _unit(1, dim, int(rng.integers(1 << 30)))[0]
            base /= np.linalg.norm(base)
            rows = rng.choice(nd, n_dup, replace=False)
            C[rows] = base[None, :] + float(rng.choice([0.0, 1e-5, 1e-4, 1e-3])) * rng.standard_normal((n_dup, dim)).astype(np.float32)
            C[rows] /= np.linalg.norm(C[rows], axis=1, keepdims=True)
        for _ in range(int(rng.integers(0, 3))):                     # a query that is an index row
            Q[int(rng.integers(nq))] = C[int(rng.integers(nd))]
    ix = HipIndex(dim, nd)
    cut = int(rng.integers(0, nd + 1))
    if cut:
        ix.add(C[:cut])
    if cut < nd:
        ix.add(C[cut:])
    ix.search_stats(reset=True)
    sc, ids = ix.search(Q, k)
    st = ix.search_stats()
    kk = min(k, nd)
    assert (ids[:, kk:] == -1).all() and np.isinf(sc[:, kk:]).all()
    _assert_ids_equal_fp64(ids[:, :kk], sc[:, :kk], C, Q, kk)
    assert st["uncertified"] == 0 and st["certified"] + st["certified_extended"] + st["flagged"] == nq, (st, nd, nq, dim, k)


@pytest.mark.parametrize("nq", [1, 40])
def test_certification_bound_holds_at_bf16_rounding_midpoints(nq):
    """bf16 keeps 8 significand bits, its unit roundoff is 2^-8 — a bound built on 2^-9 per operand
    (rounds 1-3: eps_rel = 2^-8 + 2^-18 + ...) certifies wrong results when components sit just below rounding midpoints.
    Construction (dim 256, two blocks of 128 components): the query is 1/16 everywhere, scaled by (1 + 2^-8 - 2^-20) on
    block 1 (rounds DOWN to 1/16).  Row A lives on block 1 with the same just-below-midpoint components: bf16 score 0.5,
    fp32 score 0.5 (1 + 2^-8)^2 = 0.503917.  Forty rows B live on block 2 with bf16-exact components (1 + 2^-7) / 16:
    bf16 = fp32 score 0.503906.  The fp32 ranking puts A FIRST; by bf16 score it is 3.9e-3 behind all forty — 5.5e-3
    |q| max|d|, outside the old bound's band (3.95e-3), inside the true one (7.8e-3) and inside the default bound, which
    measures the residuals of this very data."""
    dim, nd, k = 256, 3000, 10
    rng = np.random.default_rng(71)
    th = np.float32(1.0 + 2.0 ** -8 - 2.0 ** -20)
    C = (0.3 * _unit(nd, dim, 72)).astype(np.float32)                        # filler: small norms, scores ~ 0.02
    B = np.zeros(dim, np.float32); B[128:] = np.float32((1.0 + 2.0 ** -7) / 16.0)
    A = np.zeros(dim, np.float32); A[:128] = np.float32(1.0 / 16.0) * th
    C[:40] = B
    C[50] = A
    Q = (0.5 * _unit(nq, dim, 73)).astype(np.float32)
    Q[0] = np.float32(1.0 / 16.0)
    Q[0, :128] *= th
    assert A[0] != np.float32(1.0 / 16.0) and torch.tensor(A[:1]).to(torch.bfloat16).float().item() == 1.0 / 16.0    # rounds down
    ref = C.astype(np.float64) @ Q[0].astype(np.float64)
    assert int(np.argmax(ref)) == 50 and ref[50] - ref[0] > 5e-6              # the fp64 (and fp32) ranking: A first
    ix = HipIndex(dim, nd); ix.add(C)
    em = ix.error_model()
    assert 3.8e-3 * np.linalg.norm(A) < em["max_row_bf16_residual"] < 4.0e-3 * np.linalg.norm(A), em     # |A - bf16(A)| = 2^-8 |A|
    assert abs(em["worst_case_eps_rel"] - (2.0 ** -7 + 2.0 ** -16 + (2 * dim + 128) * 2.0 ** -24)) < 1e-6
    ix.search_stats(reset=True)
    sc, ids = ix.search(Q, k)
    st = ix.search_stats()
    assert ids[0, 0] == 50 and list(ids[0, 1:]) == list(range(9)), ids[0]
    _assert_ids_equal_fp64(ids, sc, C, Q, k)
    assert st["uncertified"] == 0
    # the worst-case relative bound as the caller's model: the same answer
    ix.set_search_eps(em["worst_case_eps_rel"])
    sc2, ids2 = ix.search(Q, k)
    assert np.array_equal(ids2, ids)
    # what rounds 1-3 shipped as "rigorous" (2^-9 per operand): certifies the forty B rows and never looks at A
    ix.set_search_eps(2.0 ** -8 + 2.0 ** -18 + (2 * dim + 128) * 2.0 ** -24)
    sc3, ids3 = ix.search(Q, k)
    assert 50 not in ids3[0]


@pytest.mark.parametrize("n_dup,nq", [(3000, 1), (3000, 40), (3000, 300), (9000, 40)])
def test_contiguous_near_duplicate_block_goes_through_the_band_pass(n_dup, nq):
    """A templated document (a slide deck, a form) embedded page after page: a CONTIGUOUS block of near-duplicates — all of
    it lands in a few half-lists of one chunk, which overflow, and the pre-pass threshold sits inside the block.  The merge
    cannot certify; the band pass scores the flagged queries against every row (bf16 GEMM), re-scores the whole band in fp32
    and returns the fp32 ranking — without sweeping the fp32 index once per 8 queries.  More than 8192 rows inside the band:
    the exact fp32 pass."""
    dim, nd, k = 2304, 24000, 10
    C = _unit(nd, dim, 81)
    Q = _unit(nq, dim, 82)
    rng = np.random.default_rng(83)
    base = Q[0] + 0.5 * _unit(1, dim, 84)[0]
    base /= np.linalg.norm(base)
    C[6000:6000 + n_dup] = base[None, :] + 1e-4 * rng.standard_normal((n_dup, dim)).astype(np.float32)
    C[6000:6000 + n_dup] /= np.linalg.norm(C[6000:6000 + n_dup], axis=1, keepdims=True)
    ix = HipIndex(dim, nd); ix.add(C)
    ix.search_stats(reset=True)
    sc, ids = ix.search(Q, k)
    st = ix.search_stats()
    assert set(ids[0].tolist()) <= set(range(6000, 6000 + n_dup))
    _assert_ids_equal_fp64(ids, sc, C, Q, k)
    assert st["uncertified"] == 0 and st["certified"] + st["certified_extended"] + st["flagged"] == nq, st
    if n_dup <= 8192:
        assert st["band_pass"] >= 1 and st["exact_pass"] == 0, st
    else:
        assert st["exact_pass"] >= 1, st
    # the packed-key output of the multi-GPU exchange takes the same route
    from visrag_amd.retriever import unpack_keys_host
    keys = ix.search_keys(torch.from_numpy(Q).cuda(), k, id_offset=7)
    us, ui = unpack_keys_host(keys.cpu().numpy())
    assert np.array_equal(ui, ids + 7) and np.array_equal(us, sc)


RETRIEVE_SHARD_WORKER = r'''
import os, sys, types
sys.path.insert(0, os.environ["VR_ROOT"])
import numpy as np, torch, torch.distributed as dist
from visrag_amd.retriever import distributed_parallel_retrieve
from visrag_amd.utils import save_as_trec
rank = int(os.environ["VR_RANK"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["VR_PORT"], rank=rank, world_size=2)
torch.cuda.set_device(0)
calls = {"n": 0}
real_gather = dist.all_gather_into_tensor
def counting_gather(*a, **k):
    calls["n"] += 1
    return real_gather(*a, **k)
dist.all_gather_into_tensor = counting_gather
out = os.environ["VR_OUT"]
args = types.SimpleNamespace(output_dir=out, process_index=rank, device="cuda:0")
for k in (10, 40):                                        # fused sweep and the deep path
    for gt in (False, True):
        ref = distributed_parallel_retrieve(args, k, global_topk=gt, sharded=False)      # every rank: ALL corpus shards (the reference)
        n0 = calls["n"]
        got = distributed_parallel_retrieve(args, k, global_topk=gt, sharded=True)       # rank r: only the shards rank r wrote
        assert calls["n"] == n0 + 1, calls
        assert got == ref and list(got) == list(ref), (rank, k, gt)
        save_as_trec(ref, os.path.join(out, f"ref.{rank}.{k}.{gt}.trec")); save_as_trec(got, os.path.join(out, f"got.{rank}.{k}.{gt}.trec"))
        a = open(os.path.join(out, f"ref.{rank}.{k}.{gt}.trec"), "rb").read(); b = open(os.path.join(out, f"got.{rank}.{k}.{gt}.trec"), "rb").read()
        assert a == b and len(a) > 0, (rank, k, gt)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_corpus_sharded_retrieve_two_ranks_on_one_gpu(tmp_path):
    """Row g on the device: two processes (sharing the one GPU, gloo rendezvous), each loading ONLY its own corpus shards
    into a real HipIndex behind `distributed_parallel_retrieve(sharded=True)` — equal to the replicated form (every rank
    loads everything: what the reference does) dict for dict and TREC byte for byte; union and global top-k; k = 10 (fused
    sweep) and 40 (deep path); rank 0 wrote two split files, one of rank 1's rows duplicates one of rank 0's."""
    import socket, subprocess, sys
    from visrag_amd.utils import shard_name, write_shard
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dim, nd, nq = 512, 9000, 70
    C, Q = _unit(nd, dim, 91), _unit(nq, dim, 92)
    C[8000] = C[17]
    docs = [f"doc{i}" for i in range(nd)]
    out = tmp_path / "emb"; out.mkdir()
    write_shard(str(out / shard_name("corpus", 0, 0, 3000)), C[:3000], docs[:3000])
    write_shard(str(out / shard_name("corpus", 0, 3000, 5000)), C[3000:5000], docs[3000:5000])
    write_shard(str(out / shard_name("corpus", 1)), C[5000:], docs[5000:])
    write_shard(str(out / shard_name("query", 0)), Q[:40], [f"q{i}" for i in range(40)])
    write_shard(str(out / shard_name("query", 1)), Q[40:], [f"q{i}" for i in range(40, nq)])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(RETRIEVE_SHARD_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, VR_ROOT=root, VR_PORT=str(port), VR_RANK=str(r), VR_OUT=str(out))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def _prepass_spot_rows(nd, spots):
    """rows of the owning pre-pass's sampled tiles (kernels.h: index tile k * S + S - 1, S = tiles // 16)"""
    tiles = (nd + 255) // 256
    S = tiles // 16
    return np.concatenate([np.arange((k * S + S - 1) * 256, min(nd, (k * S + S) * 256)) for k in spots])


@pytest.mark.parametrize("nd,nq,dim,k", [(33024, 300, 256, 10), (33024, 1000, 128, 26), (36700, 1000, 128, 10), (100000, 1000, 256, 10),
                                          (47000, 520, 384, 16)])
def test_prepass_that_owns_its_sample(nd, nq, dim, k):
    """Round 6 (search.hip: search_prepass_own_kernel): where it shortens the sweep, the threshold pre-pass keeps every score of
    its 16 sampled 256-row tiles and appends the rows that reach the threshold to lists of its own; the sweep skips those tiles.
    Shapes where the plan says so (incl. 36 700 rows: the last, ragged tile is a sampled one), every query's best row planted
    INSIDE a sampled tile for a third of the queries, next to one for another third: ids equal an fp64 brute force."""
    C, Q = _unit(nd, dim, 71), _unit(nq, dim, 72)
    spot = _prepass_spot_rows(nd, range(16))
    rng = np.random.default_rng(73)
    for q in range(0, nq, 3):                                   # best row inside a sampled tile
        r = int(rng.choice(spot)); C[r] = Q[q] + 0.3 * C[r]; C[r] /= np.linalg.norm(C[r])
    for q in range(1, nq, 3):                                   # ... and just outside one (the row in front of the tile / behind it)
        r = int(rng.choice([spot[0] - 1, spot[255] + 1, spot[256 * 7] - 1])); C[r] = Q[q] + 0.3 * C[r]; C[r] /= np.linalg.norm(C[r])
    ix = HipIndex(dim, nd); ix.add(C)
    plan = ix.search_plan(nq)
    assert plan["prepass_chunks"] == 8 and plan["list_chunks"] == plan["sweep_chunks"] + 8, plan
    ix.search_stats(reset=True)
    sc, ids = ix.search(Q, k)
    st = ix.search_stats()
    _assert_ids_equal_fp64(ids, sc, C, Q, k)
    assert st["uncertified"] == 0 and st["certified"] + st["certified_extended"] + st["flagged"] == nq, st
    assert st["flagged"] <= nq // 50, st                       # random rows: the lists prove the result, nothing behind the sweep
    assert ix.search_plan(16)["prepass_chunks"] == 0 and ix.search_plan(1)["prepass_chunks"] == 0
    # the packed-key output of the multi-GPU exchange takes the same route (a 2-way shard of the 100k index is such a shape)
    from visrag_amd.retriever import unpack_keys_host
    keys = ix.search_keys(torch.from_numpy(Q).cuda(), k, id_offset=11)
    us, ui = unpack_keys_host(keys.cpu().numpy())
    assert np.array_equal(ui, ids + 11) and np.array_equal(us, sc)


def test_prepass_own_lists_overflow_is_flagged_not_wrong():
    """More sampled rows reach a query's threshold than its own lists hold (1024): five sampled tiles are 1280 copies of one
    vector that is the query's best row — twenty fold groups tie at the top, the threshold IS that score, all 1280 pass.  The
    lists are then declared incomplete (thr_cert = +inf) and the band pass redoes the query: the ten LOWEST row ids of the copies."""
    nd, nq, dim, k = 33024, 300, 256, 10
    C, Q = _unit(nd, dim, 81), _unit(nq, dim, 82)
    rows = _prepass_spot_rows(nd, [2, 3, 4, 9, 15])
    assert len(rows) == 1280
    C[rows] = Q[0]
    C[777] = Q[0]                                               # one more copy in a swept tile, with a lower id than any sampled one
    ix = HipIndex(dim, nd); ix.add(C)
    assert ix.search_plan(nq)["prepass_chunks"] == 8
    ix.search_stats(reset=True)
    sc, ids = ix.search(Q, k)
    st = ix.search_stats()
    assert list(ids[0]) == [777] + rows[:9].tolist(), ids[0]
    _assert_ids_equal_fp64(ids, sc, C, Q, k)
    assert st["flagged"] >= 1 and st["uncertified"] == 0, st


def test_streaming_search_hands_huge_bands_to_the_exact_pass_after_the_first():
    """A handful of queries whose error band is the whole index (an error model nothing satisfies): the FIRST such search is
    redone by the merge workgroups themselves (no launches behind the streaming search: 70 ms per query at 100 000 rows) and
    sets the index's host-visible word; from the second search on the exact fp32 pass is launched behind the merge and takes
    the flagged queries (~1 ms per 8 queries).  Same ids and the same floats either way; vr_index_reset clears the word."""
    nd, nq, dim, k = 30000, 8, 2304, 10
    C, Q = _unit(nd, dim, 91), _unit(nq, dim, 92)
    ix = HipIndex(dim, nd); ix.add(C)
    ix.set_search_eps(100.0)
    ix.search_stats(reset=True)
    assert ix.search_plan(nq)["exact_pass_launched"] == 0
    sc1, ids1 = ix.search(Q, k)
    st1 = ix.search_stats(reset=True)
    assert st1["flagged"] == nq and st1["exact_pass"] == nq, st1
    assert ix.search_plan(nq)["exact_pass_launched"] == 1
    sc2, ids2 = ix.search(Q, k)
    st2 = ix.search_stats(reset=True)
    assert st2["flagged"] == nq and st2["exact_pass"] == nq, st2
    assert np.array_equal(ids1, ids2) and np.array_equal(sc1, sc2)
    _assert_ids_equal_fp64(ids2, sc2, C, Q, k)
    ix.set_search_eps(None)                                      # ordinary queries on the same index: nothing is flagged, results exact
    sc3, ids3 = ix.search(Q, k)
    assert np.array_equal(ids3, ids1)
    ix.reset(); ix.add(C)
    assert ix.search_plan(nq)["exact_pass_launched"] == 0
    # the sweeps behind a pre-pass likewise (band_select_kernel walks the first huge bands itself)
    Q40 = _unit(40, dim, 93)
    ix.set_search_eps(100.0)
    sc4, ids4 = ix.search(Q40, k)
    assert ix.search_plan(40)["exact_pass_launched"] == 1
    sc5, ids5 = ix.search(Q40, k)
    assert np.array_equal(ids4, ids5) and np.array_equal(sc4, sc5)
    _assert_ids_equal_fp64(ids5, sc5, C, Q40, k)
