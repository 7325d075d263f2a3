"""-m gpu: BASELINE.json configs[2] AS WORDED — "100k synthetic page corpus: embed + HBM-resident index + 1k-query
top-10" — through the drop-in entry points (reference flow: src/openmatch/inference/inference.py:53-172 ->
retriever/dense_retriever.py:13-97):

    distributed_parallel_embedding_inference(corpus of DISTINCT synthetic pages)   ->  pickle shards (split_save)
    distributed_parallel_embedding_inference(1 000 text queries)                   ->  query shard
    distributed_parallel_retrieve(args, 10, global_topk=True)                       ->  {qid: {docid: score}}

and the ids against an fp64 brute force over the SAME fp32 embeddings (the checker: torch fp64 matmul + topk).
The index is made of model-produced embeddings (page-page cosines ~0.6, query->page scores a few 1e-2 apart), not of
i.i.d. Gaussian rows: the certification statistics asserted here are statements about THAT geometry.

The corpus size is 100 000 pages (about 2.5 minutes of GPU); VISRAG_TEST_CORPUS_PAGES shrinks it for a quick look."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from visrag_amd import utils as U  # noqa: E402
from visrag_amd.config import full_config  # noqa: E402
from visrag_amd.engine import HipIndex  # noqa: E402
from visrag_amd.inference import distributed_parallel_embedding_inference  # noqa: E402
from visrag_amd.modeling import DRModelForInference  # noqa: E402
from visrag_amd.retriever import distributed_parallel_retrieve  # noqa: E402
from visrag_amd.synth import iter_synth_weights, synth_pages, synth_pages_gpu, synth_queries  # noqa: E402
from visrag_amd.tokenizer import StandInTokenizer  # noqa: E402

QUERY_PREFIX = "Represent this query for retrieving relevant documents: "
N_PAGES = int(os.environ.get("VISRAG_TEST_CORPUS_PAGES", "100000"))
N_QUERIES, K, BATCH = 1000, 10, 32


def test_gpu_page_generator_equals_the_host_generator():
    """vr_synth_pages == visrag_amd.synth.synth_pages, bit for bit (the corpus of this file is the same corpus the CPU
    oracle and the fixtures see)."""
    for size, seed, first, n in ((448, 0, 0, 3), (448, 3, 99_990, 2), (224, 1, 7, 2)):
        host = synth_pages(n, size=size, seed=seed, first=first)
        dev = synth_pages_gpu(n, size=size, seed=seed, first=first).cpu().numpy()
        assert np.array_equal(host, dev), (size, seed, first)


class _SynthCorpus:
    """Iterable of {'id', 'text', 'image'} like the reference's InferenceDataset (inference.py:40-50); the images are u8
    HWC device tensors generated a batch at a time (a fresh buffer per batch: the previous one may still be read)."""

    def __init__(self, n):
        self.n = n

    def __iter__(self):
        for lo in range(0, self.n, BATCH):
            nb = min(BATCH, self.n - lo)
            px = synth_pages_gpu(nb, size=448, seed=0, first=1_000_000 + lo)
            for i in range(nb):
                yield {"id": f"page{lo + i}", "text": "", "image": px[i]}


def test_config3_embed_100k_pages_index_and_search_ids_equal_fp64(tmp_path):
    cfg = full_config()
    model = DRModelForInference.build(cfg=cfg, state_dict=iter_synth_weights(cfg, 0, device="cuda"), max_images=BATCH,
                                      max_tokens=4096, max_seqs=64, pipeline=1)
    tok = StandInTokenizer(cfg.vocab_size)
    args = types.SimpleNamespace(output_dir=str(tmp_path), per_device_eval_batch_size=BATCH, process_index=0, world_size=1,
                                 max_inmem_docs=max(BATCH, N_PAGES // 4), device=f"cuda:{model.encoder.device}")
    distributed_parallel_embedding_inference(_SynthCorpus(N_PAGES), model, args, "corpus", True,
                                             {"tokenizer": tok, "max_inp_length": 2048})
    queries = [{"id": f"q{i}", "text": QUERY_PREFIX + t, "image": None} for i, t in enumerate(synth_queries(N_QUERIES, seed=0))]
    distributed_parallel_embedding_inference(queries, model, args, "query", False, {"tokenizer": tok, "max_inp_length": 512})
    for enc, _ in model._slots:
        enc.close()
    shards = U.list_shards(str(tmp_path), "corpus")
    assert len(shards) >= 4
    run = distributed_parallel_retrieve(args, K, global_topk=True)
    assert len(run) == N_QUERIES and all(len(v) == K for v in run.values())

    # ---- the checker: fp64 brute force over the same fp32 rows (files in the order the retriever walks them)
    P = np.concatenate([U.read_shard(p)[0] for p in shards])
    ids = [i for p in shards for i in U.read_shard(p)[1]]
    Q, qids = U.read_shard(U.list_shards(str(tmp_path), "query", 0)[0])
    assert P.shape == (N_PAGES, cfg.hidden_size) and sorted(ids) == sorted(f"page{i}" for i in range(N_PAGES)) and len(qids) == N_QUERIES
    np.testing.assert_allclose(np.linalg.norm(P, axis=1), 1.0, atol=1e-5)
    Pd, Qd = torch.from_numpy(P).cuda(), torch.from_numpy(Q).cuda()
    ref = Qd.double() @ Pd.double().T
    rv, ri = torch.topk(ref, K, dim=1)
    rv, ri = rv.cpu().numpy(), ri.cpu().numpy()
    n_same = 0
    pos = {d: j for j, d in enumerate(ids)}
    for qi, qid in enumerate(qids):
        got = sorted(run[qid].items(), key=lambda kv: -kv[1])
        got_ids = [d for d, _ in got]
        want_ids = [ids[j] for j in ri[qi]]
        if got_ids == want_ids:
            n_same += 1
        for c, (d, s) in enumerate(got):
            exact = float(ref[qi, pos[d]])
            assert abs(exact - s) < 2e-6, (qid, d)                          # returned scores are the rows' fp32 dots
            if d != want_ids[c]:
                assert abs(exact - rv[qi, c]) < 3e-7, (qid, c, d, want_ids[c], exact, rv[qi, c])    # differs only inside fp32 summation noise
    assert n_same >= N_QUERIES - 5, n_same

    # ---- what the certification did on this geometry (the same index, directly)
    ix = HipIndex(cfg.hidden_size, N_PAGES)
    ix.add(Pd)
    ix.search_stats(reset=True)
    sc, idx = ix.search(Qd, K)
    st = ix.search_stats()
    em = ix.error_model()
    assert st["uncertified"] == 0 and st["certified"] + st["certified_extended"] + st["flagged"] == N_QUERIES, st
    assert 1e-3 < em["max_row_bf16_residual"] < 2.6e-3, em                  # measured, against the worst case 2^-8 = 3.9e-3
    differ = idx.cpu().numpy() != ri
    assert (np.abs(np.take_along_axis(ref.cpu().numpy(), idx.cpu().numpy(), 1) - rv)[differ] < 3e-7).all()
    print("config3 certification on model embeddings:", st, em)
    # one query alone (the HBM-bound streaming kernel) returns the same rows
    s1, i1 = ix.search(Qd[:1], K)
    assert torch.equal(i1, idx[:1]) and torch.equal(s1, sc[:1])
    ix.close()
