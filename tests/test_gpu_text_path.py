"""-m gpu: the reference's operating points on the TOKEN side of the path, against the CPU oracle on the same
seeded inputs: the split-precision decoder pass of token-only batches (csrc/hp_text.hip), long sequences
(`p_max_len 2048`, `q_max_len 512`: visrag_scripts/eval_retriever/eval.sh:60-75), truncation at max_inp_length
(modeling_minicpmv.py:179-180), micro-batching when a batch exceeds the workspace, and the deterministic poolings
of DRModel.encode (dense_retrieval_model.py:172-220).

"Full width" below = the real MiniCPM-V-2.0 widths (ViT 1152 / 16 heads / 4304, decoder 2304 / 36 heads / 5760,
vocab 122753) with FEWER LAYERS (ViT 2, decoder 3), so that the fp32 oracle finishes in seconds on the host cores;
the 40-layer end-to-end bar is tests/test_gpu_config1.py against the reference's own fixture."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import visrag_ret_oracle as O  # noqa: E402
from visrag_amd.config import full_config, tiny_config  # noqa: E402
from visrag_amd.engine import HipEncoder  # noqa: E402
from visrag_amd.modeling import DRModelForInference  # noqa: E402
from visrag_amd.preprocess import prepare_batch  # noqa: E402
from visrag_amd.synth import synth_pages, synth_queries, synth_state_dict  # noqa: E402
from visrag_amd.tokenizer import StandInTokenizer  # noqa: E402

TOL = 1e-3


def _words(n, seed):
    rng = np.random.default_rng(seed)
    return " ".join("w%d" % int(x) for x in rng.integers(0, 50000, size=n))


def _oracle(W, cfg, items, pooling="wmean"):
    return O.encode(W, cfg, [it.input_ids for it in items], [it.image_bound for it in items], [it.slices for it in items],
                    pooling=pooling).numpy()


def _scores_err(a, ref, others):
    """largest error of the scores of `a` rows against a set of unit vectors, vs the reference rows'"""
    return float(np.abs(a @ others.T - ref @ others.T).max())


@pytest.fixture(scope="module")
def wide_model():
    cfg = full_config()
    cfg.vit_depth, cfg.num_layers = 2, 3
    W = synth_state_dict(cfg, 0)
    # weights that are NOT bf16-representable (the synthetic checkpoint is): the low halves of the weight split matter
    for k in list(W.keys()):
        if k.startswith("llm.model.layers.") and k.endswith("proj.weight"):
            W[k] = (W[k] * 1.0009765625).contiguous()
    W["llm.model.embed_tokens.weight"] = (W["llm.model.embed_tokens.weight"] * 0.99951171875).contiguous()
    enc = HipEncoder(cfg, max_images=4, max_tokens=4096, max_seqs=32)
    enc.load_state_dict(((k, v.cuda()) for k, v in W.items()))
    model = DRModelForInference(cfg, enc, gpu_preprocess=False)
    yield cfg, W, enc, model
    enc.close()


def test_text_queries_split_precision_full_width(wide_model):
    """16 queries (~20 tokens) through the split-precision pass: fp32-class agreement with the oracle — 1 - cos < 1e-6
    and score errors below 1e-4 against random unit directions (the bf16 pass sits at ~1e-3 here, the bar)."""
    cfg, W, enc, model = wide_model
    tok = StandInTokenizer(cfg.vocab_size)
    texts = ["Represent this query for retrieving relevant documents: " + t for t in synth_queries(16, seed=0)]
    items = prepare_batch(texts, [None] * 16, tok, cfg, 512)
    ref = _oracle(W, cfg, items)
    got = model.encode_prepared(items).cpu().numpy()
    np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    cos = (got * ref).sum(1)
    assert cos.min() > 1 - 1e-6, 1 - cos.min()
    rng = np.random.default_rng(1)
    U = rng.standard_normal((256, cfg.hidden_size)).astype(np.float32)
    U /= np.linalg.norm(U, axis=1, keepdims=True)
    assert _scores_err(got, ref, np.concatenate([U, ref])) < 1e-4


def test_single_query_takes_the_weight_streamer_full_width(wide_model):
    """A batch of at most 32 tokens (one query: the serving case, BASELINE config 5's retrieve step) runs the split-precision
    decoder pass on the weight-streaming GEMM (gemm_skinny.hip: every projection = ONE pass over its weights per operand
    half, fp32 split-K planes summed in a fixed order) instead of 128-row tiles: same fp32-class agreement with the
    oracle, and the embedding of a query alone equals its embedding inside a batch of 16 to fp32 noise."""
    cfg, W, enc, model = wide_model
    tok = StandInTokenizer(cfg.vocab_size)
    texts = ["Represent this query for retrieving relevant documents: " + t for t in synth_queries(16, seed=0)]
    items = prepare_batch(texts, [None] * 16, tok, cfg, 512)
    assert all(len(it.input_ids) <= 32 for it in items) and sum(len(it.input_ids) for it in items) > 32
    batch = model.encode_prepared(items).cpu().numpy()
    alone = np.concatenate([model.encode_prepared(items[i:i + 1]).cpu().numpy() for i in (0, 5, 15)])
    ref = _oracle(W, cfg, [items[i] for i in (0, 5, 15)])
    assert ((alone * ref).sum(1)).min() > 1 - 1e-6, 1 - (alone * ref).sum(1).min()
    np.testing.assert_allclose(alone, batch[[0, 5, 15]], atol=3e-6, rtol=0)
    # three short queries in ONE call (22 rows together: still the streamer) == each alone, and a 31-token one
    short = prepare_batch(synth_queries(3, seed=7, min_words=4, max_words=8), [None] * 3, tok, cfg, 512)
    assert sum(len(it.input_ids) for it in short) <= 32
    together = model.encode_prepared(short).cpu().numpy()
    each = np.concatenate([model.encode_prepared(short[i:i + 1]).cpu().numpy() for i in range(3)])
    np.testing.assert_allclose(together, each, atol=3e-6, rtol=0)
    assert ((together * _oracle(W, cfg, short)).sum(1)).min() > 1 - 1e-6
    long_q = prepare_batch([_words(30, 11)], [None], tok, cfg, 512)
    assert len(long_q[0].input_ids) == 31
    got = model.encode_prepared(long_q).cpu().numpy()
    assert ((got * _oracle(W, cfg, long_q)).sum(1)).min() > 1 - 1e-6


def test_long_passage_and_long_query_full_width(wide_model):
    """The reference's maximum lengths: one 2048-token text passage and one 512-token query (both LONGER texts,
    truncated at max_inp_length like modeling_minicpmv.py:179-180), next to a short one in the same batch."""
    cfg, W, enc, model = wide_model
    tok = StandInTokenizer(cfg.vocab_size)
    for max_len, n_words in ((2048, 2600), (512, 700)):
        texts = [_words(n_words, 3), "short text", _words(max_len - 1, 4)]          # the last one: exactly max_len with bos
        items = prepare_batch(texts, [None] * 3, tok, cfg, max_len)
        assert [len(it.input_ids) for it in items] == [max_len, 3, max_len]
        ref = _oracle(W, cfg, items)
        got = model.encode_prepared(items).cpu().numpy()
        cos = (got * ref).sum(1)
        assert cos.min() > 1 - 1e-6, (max_len, 1 - cos.min())
        # the truncated item equals the same text cut by hand
        cut = prepare_batch([" ".join(texts[0].split()[: max_len - 1])], [None], tok, cfg, None)
        assert cut[0].input_ids == items[0].input_ids


def test_page_with_long_caption_bf16_path_full_width(wide_model):
    """Image + ~1900 caption tokens in ONE item (L ~ 2000: the bf16 flash-attention path at the reference's p_max_len),
    a plain page beside it, and the same item truncated mid-caption at max_inp_length = 1024."""
    from PIL import Image
    cfg, W, enc, model = wide_model
    tok = StandInTokenizer(cfg.vocab_size)
    pages = synth_pages(2, size=448, seed=2)
    texts = [_words(1900, 5), ""]
    for max_len in (2048, 1024):
        items = prepare_batch(texts, [Image.fromarray(p) for p in pages], tok, cfg, max_len)
        assert len(items[0].input_ids) == min(max_len, 68 + 1900) and len(items[1].input_ids) == 68
        ref = _oracle(W, cfg, items)
        got = model.encode_prepared(items).cpu().numpy()
        cos = (got * ref).sum(1)
        assert cos.min() > 1 - TOL, (max_len, 1 - cos.min())
    with pytest.raises(ValueError):                     # cut inside the image placeholder: the reference's hstack fails too
        prepare_batch(texts, [Image.fromarray(p) for p in pages], tok, cfg, 40)


def test_micro_batch_split_when_a_batch_exceeds_the_workspace(wide_model):
    """24 items x ~300 tokens = 7200 tokens against max_tokens = 4096: encode_prepared cuts the batch into
    micro-batches; results equal item-by-item encoding; one item beyond the workspace is refused loudly."""
    cfg, W, enc, model = wide_model
    tok = StandInTokenizer(cfg.vocab_size)
    texts = [_words(299, 100 + i) for i in range(24)]
    items = prepare_batch(texts, [None] * 24, tok, cfg, 2048)
    assert sum(len(it.input_ids) for it in items) > enc.max_tokens
    got = model.encode_prepared(items).cpu().numpy()
    ref = _oracle(W, cfg, items[:4])
    assert ((got[:4] * ref).sum(1)).min() > 1 - 1e-6
    one = np.concatenate([model.encode_prepared(items[i:i + 1]).cpu().numpy() for i in (0, 13, 23)])
    np.testing.assert_allclose(got[[0, 13, 23]], one, atol=2e-6)
    too_long = prepare_batch([_words(5000, 7)], [None], tok, cfg, None)
    with pytest.raises(ValueError):
        model.encode_prepared(too_long)


def test_an_items_precision_route_does_not_depend_on_its_batch_mates(wide_model):
    """ADVICE r4 #5 / verdict r5 item 9: the split-precision route is chosen per ITEM.  The same queries (a) in a token-only
    call, (b) in a call that also holds pages, (c) one at a time: (b) is bit-identical to (a) — the text items of a mixed call
    form the same micro-batch —, (c) agrees at fp32 class (a lone query runs the weight-streaming GEMMs: another summation
    order, nothing else), all three at 1 - cos < 1e-6 of the fp32 oracle; the pages beside them are bit-identical to the pages
    alone; and through the reference-shaped call `model(passage={text, image})` with a None image among the pages."""
    from PIL import Image
    cfg, W, enc, model = wide_model
    tok = StandInTokenizer(cfg.vocab_size)
    qtexts = ["Represent this query for retrieving relevant documents: " + q for q in synth_queries(5, seed=3)]
    pages = [Image.fromarray(p) for p in synth_pages(2, size=448, seed=4)]
    q_items = prepare_batch(qtexts, [None] * 5, tok, cfg, 512)
    p_items = prepare_batch(["", ""], pages, tok, cfg, 2048)
    ref_q = _oracle(W, cfg, q_items)
    a = model.encode_prepared(q_items).cpu().numpy()
    mixed = [q_items[0], p_items[0], q_items[1], q_items[2], p_items[1], q_items[3], q_items[4]]
    b_all = model.encode_prepared(mixed).cpu().numpy()
    b = b_all[[0, 2, 3, 5, 6]]
    c = np.concatenate([model.encode_prepared([it]).cpu().numpy() for it in q_items])
    assert np.array_equal(a, b)                                              # identical bits beside a page
    assert np.abs(a - c).max() < 3e-6 and ((a * c).sum(1)).min() > 1 - 1e-6  # alone: the streaming GEMMs' summation order
    for x in (a, b, c):
        assert ((x * ref_q).sum(1)).min() > 1 - 1e-6
    p_alone = model.encode_prepared(p_items).cpu().numpy()
    assert np.array_equal(b_all[[1, 4]], p_alone)
    out = model(passage={"id": list("abc"), "text": [qtexts[0], "", qtexts[1]], "image": [None, pages[0], None]}, tokenizer=tok,
                max_inp_length=2048).p_reps.cpu().numpy()
    assert ((out[[0, 2]] * ref_q[:2]).sum(1)).min() > 1 - 1e-6
    assert np.abs(out[1] - p_alone[0]).max() < 2e-6 or ((out[1] * p_alone[0]).sum()) > 1 - 1e-5      # (one page instead of two in the call)


def test_split_precision_beats_the_bf16_pass_and_can_be_switched_off():
    """Same tiny model with text_split_precision on / off against the oracle: both inside the tolerance the tiny
    fixtures use, the split pass two orders of magnitude closer."""
    errs = {}
    for on in (True, False):
        cfg = tiny_config()
        cfg.text_split_precision = on
        W = synth_state_dict(cfg, 0)
        enc = HipEncoder(cfg, max_images=2, max_tokens=1024, max_seqs=16)
        enc.load_state_dict(((k, v.cuda()) for k, v in W.items()))
        model = DRModelForInference(cfg, enc)
        tok = StandInTokenizer(cfg.vocab_size)
        items = prepare_batch(synth_queries(8, seed=3) + [_words(300, 9)], [None] * 9, tok, cfg, 512)
        ref = _oracle(W, cfg, items)
        got = model.encode_prepared(items).cpu().numpy()
        errs[on] = float(np.abs(got - ref).max())
        assert ((got * ref).sum(1)).min() > 1 - TOL
        enc.close()
    assert errs[True] < 2e-5 and errs[True] < 0.05 * errs[False], errs


@pytest.mark.parametrize("pooling", ["wmean", "mean", "lasttoken", "cls"])
def test_poolings_vs_oracle(pooling):
    """DRModel.encode's deterministic poolings on a mixed batch (pages, a sliced page, queries of different lengths)."""
    from PIL import Image
    cfg = tiny_config()
    W = synth_state_dict(cfg, 0)
    enc = HipEncoder(cfg, max_images=8, max_tokens=2048, max_seqs=16)
    enc.load_state_dict(((k, v.cuda()) for k, v in W.items()))
    model = DRModelForInference(cfg, enc, pooling=pooling, gpu_preprocess=False)
    tok = StandInTokenizer(cfg.vocab_size)
    pages = [p for p in synth_pages(2, size=cfg.scale_resolution, seed=0)] + [synth_pages(1, size=300, seed=5)[0][:200, :300]]
    items = prepare_batch(["", "a caption", ""], [Image.fromarray(p) for p in pages], tok, cfg, 2048)
    ref = _oracle(W, cfg, items, pooling)
    got = model.encode_prepared(items).cpu().numpy()
    # (lasttoken / cls are ONE token's state on the bf16 pass of a 256-wide model: no averaging over the sequence)
    assert ((got * ref).sum(1)).min() > 1 - (TOL if pooling == "wmean" else 5 * TOL), (pooling, (got * ref).sum(1))
    qitems = prepare_batch(synth_queries(5, seed=1) + ["x"], [None] * 6, tok, cfg, 512)
    qref = _oracle(W, cfg, qitems, pooling)
    qgot = model.encode_prepared(qitems).cpu().numpy()
    assert ((qgot * qref).sum(1)).min() > 1 - 1e-6, pooling
    for bad in ("drop_wmean", "drop_mean", "lasttoken_simcse", "bogus"):
        with pytest.raises(ValueError):
            DRModelForInference(cfg, enc, pooling=bad)
    enc.close()
