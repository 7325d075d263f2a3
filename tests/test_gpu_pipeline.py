"""-m gpu: the reference's encode -> pickle shards -> retrieve -> TREC pipeline driven through the
drop-in entry points (inference.distributed_parallel_embedding_inference,
retriever.distributed_parallel_retrieve, utils.save_as_trec) on the tiny-dims model, against the
oracle's embeddings + retrieval on the same inputs (config 1 of BASELINE.json in miniature:
pages + text queries, brute-force cosine top-3)."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import visrag_ret_oracle as O  # noqa: E402
from visrag_amd import utils as U  # noqa: E402
from visrag_amd.config import tiny_config  # noqa: E402
from visrag_amd.engine import HipEncoder  # noqa: E402
from visrag_amd.inference import distributed_parallel_embedding_inference  # noqa: E402
from visrag_amd.modeling import DRModelForInference, encode  # noqa: E402
from visrag_amd.preprocess import prepare_batch  # noqa: E402
from visrag_amd.retriever import distributed_parallel_retrieve  # noqa: E402
from visrag_amd.synth import iter_synth_weights, synth_pages, synth_queries, synth_state_dict  # noqa: E402
from visrag_amd.tokenizer import StandInTokenizer  # noqa: E402

QUERY_PREFIX = "Represent this query for retrieving relevant documents: "


def test_encode_retrieve_trec_pipeline(tmp_path):
    from PIL import Image
    cfg = tiny_config()
    enc = HipEncoder(cfg, max_images=8, max_tokens=1024, max_seqs=8)
    enc.load_state_dict(iter_synth_weights(cfg, 0, device="cuda"))
    model = DRModelForInference(cfg, enc)
    tok = StandInTokenizer(cfg.vocab_size)
    n_pages, n_q, k = 24, 6, 3
    pages = synth_pages(n_pages, size=cfg.scale_resolution, seed=3)
    qtexts = [QUERY_PREFIX + q for q in synth_queries(n_q, seed=3)]
    corpus = [{"id": f"d{i}", "text": "", "image": Image.fromarray(p)} for i, p in enumerate(pages)]
    queries = [{"id": f"q{i}", "text": t, "image": None} for i, t in enumerate(qtexts)]
    args = types.SimpleNamespace(output_dir=str(tmp_path), per_device_eval_batch_size=5, process_index=0,
                                 world_size=1, max_inmem_docs=10, device="cuda:0")
    extra = {"tokenizer": tok, "max_inp_length": 2048}
    distributed_parallel_embedding_inference(corpus, model, args, "corpus", True, extra)
    distributed_parallel_embedding_inference(queries, model, args, "query", False, extra)
    names = sorted(os.listdir(tmp_path))
    assert "embeddings.query.rank.0" in names and sum(n.startswith("embeddings.corpus.rank.0.") for n in names) >= 2
    reps, ids = U.read_shard(os.path.join(tmp_path, [n for n in names if n.startswith("embeddings.corpus")][0]))
    assert reps.dtype == np.float32 and reps.shape[1] == cfg.hidden_size and len(ids) == len(reps)

    union = distributed_parallel_retrieve(args, k)                      # reference semantics (default): k per shard
    run = distributed_parallel_retrieve(args, k, global_topk=True)      # one search over the concatenated index
    assert all(len(v) == k for v in run.values()) and all(len(v) >= k for v in union.values())
    U.save_as_trec(run, os.path.join(tmp_path, "trec", "test.0.trec"))
    assert U.load_from_trec(os.path.join(tmp_path, "trec", "test.0.trec")).keys() == run.keys()

    # ---- oracle on the same inputs
    W = synth_state_dict(cfg, 0)
    pit = prepare_batch([""] * n_pages, [c["image"] for c in corpus], tok, cfg, 2048)
    qit = prepare_batch(qtexts, [None] * n_q, tok, cfg, 2048)
    P = O.encode(W, cfg, [i.input_ids for i in pit], [i.image_bound for i in pit], [i.slices for i in pit]).numpy()
    Q = O.encode(W, cfg, [i.input_ids for i in qit], [[]] * n_q, [[]] * n_q).numpy()
    ref = O.retrieve(Q, [f"q{i}" for i in range(n_q)], [(P, [f"d{i}" for i in range(n_pages)])], n_pages)
    tol = 2.5e-3                       # 256-d tiny model (1e-3 is asserted at full dims)
    for q, docs in run.items():
        ranked = sorted(ref[q].items(), key=lambda kv: -kv[1])
        for d, s in docs.items():                                   # scores agree with the oracle
            assert abs(ref[q][d] - s) < tol, (q, d, ref[q][d], s)
        gap = ranked[k - 1][1] - ranked[k][1]
        if gap > 2 * tol:                                            # top-k set identical when well separated
            assert set(docs) == {d for d, _ in ranked[:k]}, (q, docs, ranked[:k + 1])
        for d in docs:                                               # always: nothing outside the tolerance band
            assert ref[q][d] >= ranked[k - 1][1] - 2 * tol
    # the union result contains the global top-k
    for q in run:
        assert set(run[q]) <= set(union[q])
    # demo helper signature (visrag_scripts/demo/visrag_pipeline/utils.py:12-32)
    e = encode(model, tok, [c["image"] for c in corpus[:2]])
    assert e.shape == (2, cfg.hidden_size) and np.allclose(np.linalg.norm(e, axis=1), 1, atol=1e-5)
    e = encode(model, tok, ["a text query"])
    assert e.shape == (1, cfg.hidden_size)


def test_two_batches_in_flight_give_the_same_embeddings():
    """DRModelForInference(pipeline=2): consecutive calls rotate over two workspaces / streams that
    share the weights (vr_model_clone); results are bit-identical to the single-stream model."""
    import numpy as np
    import torch
    from PIL import Image
    from visrag_amd.config import tiny_config
    from visrag_amd.modeling import DRModelForInference
    from visrag_amd.synth import iter_synth_weights
    from visrag_amd.tokenizer import StandInTokenizer
    cfg = tiny_config()
    tok = StandInTokenizer(cfg.vocab_size)
    m = DRModelForInference.build(cfg=cfg, state_dict=iter_synth_weights(cfg, 0, device="cuda"), max_images=16)
    rng = np.random.default_rng(3)
    batches = []
    for b in range(5):
        imgs = [Image.fromarray(rng.integers(0, 256, size=hw + (3,), dtype=np.uint8)) for hw in [(112, 112), (150 + 10 * b, 260)]]
        batches.append({"text": ["", f"caption {b}"], "image": imgs})
    ref = [m(passage=b, tokenizer=tok).p_reps.clone() for b in batches]
    m.set_pipeline(2)
    got = [m(passage=b, tokenizer=tok).p_reps for b in batches]      # no sync in between: two in flight
    torch.cuda.synchronize()
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    q = m(query={"text": ["what is shown"], "image": [None]}, tokenizer=tok).q_reps
    assert q.shape == (1, cfg.hidden_size) and bool(torch.isfinite(q).all())


def test_demo_build_index_and_retrieve(tmp_path):
    """visrag_amd.demo (drop-in for visrag_scripts/demo/visrag_pipeline/{build_index,answer,utils}.py): the
    knowledge base files have the reference's layout, embeddings match the oracle, retrieve() returns the
    oracle's ranking.  Page sizes are the tiny-model analogues of the reference's two demo images
    (test_image/dog.jpg 1072x670 -> 2x2 slices, cat.jpeg 2160x1790 -> 3x3)."""
    from PIL import Image
    from visrag_amd import demo
    cfg = tiny_config()
    enc = HipEncoder(cfg, max_images=12, max_tokens=2048, max_seqs=16)
    enc.load_state_dict(iter_synth_weights(cfg, 0, device="cuda"))
    model = DRModelForInference(cfg, enc)
    tok = StandInTokenizer(cfg.vocab_size)
    big = synth_pages(2, size=560, seed=11)
    pages = [Image.fromarray(p) for p in synth_pages(5, size=cfg.scale_resolution, seed=12)]
    pages.append(Image.fromarray(np.ascontiguousarray(big[0][:168, :268])))     # 268x168 = (1072x670)/4 -> 2x2
    pages.append(Image.fromarray(np.ascontiguousarray(big[1][:448, :540])))     # 540x448 = (2160x1790)/4 -> 3x3
    from visrag_amd.preprocess import slice_image
    assert slice_image(pages[5], cfg.max_slice_nums, cfg.scale_resolution, cfg.patch_size)[2] == [2, 2]
    assert slice_image(pages[6], cfg.max_slice_nums, cfg.scale_resolution, cfg.patch_size)[2] == [3, 3]
    kb = str(tmp_path / "kb")
    reps = demo.add_pages(model, tok, pages, kb, names=[f"doc.pdf_{i}.png" for i in range(len(pages))], batch_size=3)
    assert sorted(os.listdir(kb)) == sorted(["reps.npy", "index2img_filename.txt"] + [f"doc.pdf_{i}.png" for i in range(7)])
    on_disk = np.load(os.path.join(kb, "reps.npy"))
    assert on_disk.dtype == np.float32 and on_disk.shape == (7, cfg.hidden_size) and np.array_equal(on_disk, reps)
    assert open(os.path.join(kb, "index2img_filename.txt")).read().split("\n") == [f"doc.pdf_{i}.png" for i in range(7)]
    # oracle on the same pages / query
    W = synth_state_dict(cfg, 0)
    pit = prepare_batch([""] * 7, pages, tok, cfg, 2048)
    P = O.encode(W, cfg, [i.input_ids for i in pit], [i.image_bound for i in pit], [i.slices for i in pit]).numpy()
    assert ((P * reps).sum(1)).min() > 1 - 1e-3
    query = "annual revenue growth chart"
    qit = prepare_batch([demo.QUERY_INSTRUCTION + query], [None], tok, cfg, 2048)
    Q = O.encode(W, cfg, [qit[0].input_ids], [[]], [[]]).numpy()
    ref = (Q @ P.T)[0]
    paths, scores = demo.retrieve(kb, query, 3, model, tok, return_scores=True)
    assert len(paths) == 3 and all(os.path.exists(p) for p in paths)
    order = np.argsort(-ref)
    for p, s in zip(paths, scores):
        i = int(os.path.basename(p).split("_")[1].split(".")[0])
        assert abs(ref[i] - s) < 2.5e-3
        assert ref[i] >= ref[order[2]] - 5e-3
    if ref[order[2]] - ref[order[3]] > 5e-3:
        assert [int(os.path.basename(p).split("_")[1].split(".")[0]) for p in paths] == list(order[:3])
    ix, names = demo.load_knowledge_base(kb)
    assert demo.retrieve(kb, query, 3, model, tok, index=ix, names=names) == paths
    assert demo.retrieve(str(tmp_path / "missing"), query, 3, model, tok) is None
    ix.close(); enc.close()
