"""TEST INFRASTRUCTURE ONLY — generate tests/golden/*.npz by running the REFERENCE's own
code (imported from /root/reference through oracle/ref_harness.py) on CPU fp32 with the
deterministic synthetic checkpoint (visrag_amd/synth.py) and stand-in tokenizer.

    python oracle/gen_golden.py            # tiny-dims fixtures (seconds)
    python oracle/gen_golden.py --full     # + full MiniCPM-V-2.0 dims, 2 pages + 2 queries (minutes)
    python oracle/gen_golden.py --config1  # ONLY BASELINE config 1: full dims, 64 pages 448^2 (bs 16) + 16 queries
                                           # through the reference encode + retrieve (top-3), ~10 min on 8 cores

Runs only in the build container (the reference tree does not travel to the GPU box);
the resulting fixtures are committed.
"""
from __future__ import annotations

import argparse
import os
import pickle
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402
from visrag_amd.config import full_config, tiny_config  # noqa: E402
from visrag_amd.synth import synth_pages, synth_queries, synth_state_dict  # noqa: E402
from visrag_amd.tokenizer import StandInTokenizer  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
QUERY_PREFIX = "Represent this query for retrieving relevant documents: "  # eval.sh:45


def _hook_taps(model, taps):
    def tap(name, pick=lambda o: o):
        def hook(mod, inp, out):          # must return None (a value would replace the output)
            if name not in taps:
                taps[name] = pick(out).detach().clone()
        return hook
    return [
        model.vpm.blocks[0].register_forward_hook(tap("vit_block0")),
        model.vpm.norm.register_forward_hook(tap("vit_out")),
        model.resampler.register_forward_hook(tap("resampler_out")),
        model.llm.model.layers[0].register_forward_hook(tap("dec_layer0", lambda o: o[0])),
        model.llm.model.norm.register_forward_hook(tap("last_hidden")),
    ]


def run_reference(cfg, page_arrays, queries, seed=0, want_taps=True):
    from PIL import Image
    sd = synth_state_dict(cfg, seed)
    dr = ref_harness.build_reference_dr_model(cfg, sd)
    tok = StandInTokenizer(cfg.vocab_size)
    out = {}
    taps = {}
    hooks = _hook_taps(dr.lm_q, taps) if want_taps else []
    imgs = [Image.fromarray(a) for a in page_arrays]
    with torch.no_grad():
        # tokenisation as the reference does it (for the host-logic fixture)
        ctx = [dr.lm_q.prepare_context((t, im), tok) for t, im in zip([""] * len(imgs), imgs)]
        mi = dr.lm_q._process_list(tok, [c[0] for c in ctx], 2048, padding_side="right")
        out["page_input_ids"] = mi["input_ids"].numpy().astype(np.int32)
        out["page_attention_mask"] = mi["attention_mask"].numpy().astype(np.int8)
        out["page_image_bound"] = np.array(
            [np.pad(b.numpy().reshape(-1, 2), ((0, 16 - len(b)), (0, 0)), constant_values=-1)
             for b in mi["image_bound"]], dtype=np.int32)
        out["page_n_slices"] = np.array([len(c[1]) for c in ctx], dtype=np.int32)
        out["page_slice_sizes"] = np.array(
            [[(s.size if k < len(c[1]) else (-1, -1)) for k, s in
              enumerate(list(c[1]) + [None] * (10 - len(c[1])))][:10] for c in ctx], dtype=np.int32)
        p = dr(passage={"id": [str(i) for i in range(len(imgs))], "text": [""] * len(imgs),
                        "image": imgs}, tokenizer=tok, max_inp_length=2048)
        out["p_reps"] = p.p_reps.numpy()
        for h in hooks:
            h.remove()
        for k, v in taps.items():
            out["tap_" + k] = v.numpy()
        qtext = [QUERY_PREFIX + q for q in queries]
        q = dr(query={"id": [str(i) for i in range(len(qtext))], "text": qtext,
                      "image": [None] * len(qtext)}, tokenizer=tok, max_inp_length=512)
        out["q_reps"] = q.q_reps.numpy()
        mq = dr.lm_q._process_list(tok, qtext, 512, padding_side="right")
        out["query_input_ids"] = mq["input_ids"].numpy().astype(np.int32)
        out["query_attention_mask"] = mq["attention_mask"].numpy().astype(np.int8)
    return out, qtext


def reference_retrieve(p_reps, q_reps, n_shards, k):
    """Write reference-format pickle shards (inference.py:114-164) and run the reference's
    distributed_parallel_retrieve + save_as_trec on them."""
    ref_harness.install_shims()
    from openmatch.retriever.dense_retriever import distributed_parallel_retrieve
    from openmatch.utils import save_as_trec
    with tempfile.TemporaryDirectory() as d:
        n = len(p_reps)
        per = (n + n_shards - 1) // n_shards
        for s in range(n_shards):
            lo, hi = s * per, min(n, (s + 1) * per)
            with open(os.path.join(d, f"embeddings.corpus.rank.0.{lo}-{hi}"), "wb") as f:
                pickle.dump((p_reps[lo:hi], [f"doc{j}" for j in range(lo, hi)]), f, protocol=4)
        with open(os.path.join(d, "embeddings.query.rank.0"), "wb") as f:
            pickle.dump((q_reps, [f"q{j}" for j in range(len(q_reps))]), f, protocol=4)
        args = types.SimpleNamespace(output_dir=d, process_index=0, device="cpu")
        res = distributed_parallel_retrieve(args, k)
        trec = os.path.join(d, "out", "test.0.trec")
        save_as_trec(res, trec)
        with open(trec) as f:
            trec_text = f.read()
    return res, trec_text


def config1(n_pages=64, n_queries=16, bs=16, k=3):
    """BASELINE.json configs[0]: 64 page images (448x448) + 16 text queries through the reference's
    DRModelForInference in batches of 16 (README.md:146), CPU fp32, then the reference's
    distributed_parallel_retrieve top-3 over the pickle shards (4 shards of 16 pages).  The fixture
    keeps the embeddings, the full 16 x 64 score matrix, the ranked top-(k+1) and the rank-k / rank-(k+1)
    gap per query (identical top-k is asserted where the gap exceeds 2x tolerance)."""
    import time
    from PIL import Image
    cfg = full_config()
    sd = synth_state_dict(cfg, 0)
    dr = ref_harness.build_reference_dr_model(cfg, sd)
    tok = StandInTokenizer(cfg.vocab_size)
    pages = synth_pages(n_pages, size=448, seed=0)
    queries = [QUERY_PREFIX + q for q in synth_queries(n_queries, seed=0)]
    P, t0 = [], time.time()
    with torch.no_grad():
        for lo in range(0, n_pages, bs):
            imgs = [Image.fromarray(a) for a in pages[lo:lo + bs]]
            o = dr(passage={"id": [str(i) for i in range(lo, lo + len(imgs))], "text": [""] * len(imgs), "image": imgs},
                   tokenizer=tok, max_inp_length=2048)
            P.append(o.p_reps.numpy())
            print(f"pages {lo + len(imgs)}/{n_pages}  {time.time() - t0:.0f}s", flush=True)
        t_pages = time.time() - t0
        t1 = time.time()
        o = dr(query={"id": [str(i) for i in range(n_queries)], "text": queries, "image": [None] * n_queries},
               tokenizer=tok, max_inp_length=512)
        t_q = time.time() - t1
    P = np.concatenate(P).astype(np.float32)
    Q = o.q_reps.numpy().astype(np.float32)
    res, trec = reference_retrieve(P, Q, n_shards=4, k=k)          # union of per-shard top-k (reference semantics)
    S = Q @ P.T
    order = np.argsort(-S, axis=1, kind="stable")[:, :k + 1]
    top_scores = np.take_along_axis(S, order, 1)
    gap = top_scores[:, k - 1] - top_scores[:, k]
    # the reference's result must contain the brute-force top-k with the same scores
    for qi in range(n_queries):
        for j in range(k):
            assert abs(res[f"q{qi}"][f"doc{order[qi, j]}"] - top_scores[qi, j]) < 1e-6
    np.savez_compressed(os.path.join(GOLD, "config1_full.npz"), p_reps=P, q_reps=Q, scores=S.astype(np.float32),
                        top_ids=order.astype(np.int32), top_scores=top_scores.astype(np.float32), gap=gap.astype(np.float32),
                        trec=np.array(trec), n_pages=n_pages, n_queries=n_queries, k=k, page_seed=0, query_seed=0,
                        ref_seconds=np.array([t_pages, t_q], dtype=np.float32), ref_threads=torch.get_num_threads())
    print("config1_full: pages/s", n_pages / t_pages, "queries/s", n_queries / t_q, "min gap", float(gap.min()),
          "median gap", float(np.median(gap)), "threads", torch.get_num_threads())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--config1", action="store_true")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_grad_enabled(False)
    if a.config1:
        return config1()

    # ---- tiny dims: 4 single-slice pages (112x112), 2 sliced pages, 3 queries ----------
    cfg = tiny_config()
    pages = [p for p in synth_pages(4, size=cfg.scale_resolution, seed=0)]
    pages.append(synth_pages(1, size=300, seed=5)[0][:200, :300])     # 300x200 -> slices
    pages.append(synth_pages(1, size=300, seed=6)[0][:280, :126])     # tall page
    queries = synth_queries(3, seed=0)
    out, _ = run_reference(cfg, pages, queries, seed=0)
    for k in ("tap_dec_layer0", "tap_last_hidden"):      # item 0 only (68 valid tokens)
        out[k] = out[k][:1, :68]
    np.savez_compressed(os.path.join(GOLD, "tiny_encode.npz"), **out)
    print("tiny_encode:", {k: v.shape for k, v in out.items()})

    # ---- retrieval: reference retriever on a synthetic unit-norm index ------------------
    rng = np.random.default_rng(0)
    C = rng.standard_normal((600, 64)).astype(np.float32)
    C /= np.linalg.norm(C, axis=1, keepdims=True)
    Q = rng.standard_normal((7, 64)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    res, trec = reference_retrieve(C, Q, n_shards=3, k=5)
    qids = sorted(res)
    docs = np.array([[sorted(res[q], key=lambda d_: -res[q][d_])[j] for j in range(15)] for q in qids])
    scores = np.array([[res[q][d_] for d_ in row] for q, row in zip(qids, docs)], dtype=np.float32)
    np.savez_compressed(os.path.join(GOLD, "retrieve.npz"), C=C, Q=Q, qids=np.array(qids),
                        docs=docs, scores=scores, trec=np.array(trec))
    print("retrieve:", docs.shape, scores.shape, len(trec))

    if a.full:
        cfg = full_config()
        pages = [p for p in synth_pages(2, size=448, seed=0)]
        queries = synth_queries(2, seed=0)
        out, _ = run_reference(cfg, pages, queries, seed=0)
        keep = {k: v for k, v in out.items() if not k.startswith("tap_")}
        for k, v in out.items():          # taps: first page only, a strided row subset
            if k.startswith("tap_"):
                keep[k] = v[0, :: max(1, v.shape[1] // 16)].astype(np.float32)
        np.savez_compressed(os.path.join(GOLD, "full_encode.npz"), **keep)
        print("full_encode:", {k: v.shape for k, v in keep.items()})


if __name__ == "__main__":
    main()
