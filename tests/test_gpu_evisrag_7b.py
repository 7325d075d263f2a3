"""-m gpu: the EVisRAG generator at the SHAPE BASELINE config 5 names (Qwen2.5-VL-7B: hidden 3584, 28 query / 4 KV
heads of head_dim 128, intermediate 18944, vocabulary 152064, multimodal RoPE 16/24/24; tower 1280 / 16 heads of
head_dim 80 / 3420) with TWO layers / blocks and seeded synthetic weights, against the oracles
(oracle/qwen_gen_oracle.py, oracle/qwen_vision_oracle.py — pinned to HuggingFace's implementation by the tiny
fixtures) run on the host cores on the same inputs.  What is shape-dependent in the product and never exercised by
the 256-wide fixtures: the skinny GEMM's K-split table, the two-stage sampler over 152k logits, the decode attention
cut into 16 KV ranges at a 1405-row cache, the 28:4 grouped-query layout, the tower's head_dim-80 attention and its
tile-quantised GEMMs on a >= 5k-row page."""
import numpy as np
import pytest
import torch

from oracle.qwen_gen_oracle import QwenGenConfig, QwenGenOracle, apply_repetition_penalty, synth_weights
from oracle.qwen_vision_oracle import QwenVisionConfig, QwenVisionOracle, synth_vision_weights

pytestmark = pytest.mark.gpu

N_IMG, GRID = 5, (16, 16)            # five pages of 16 x 16 image tokens = 1280 tokens (bench_generate's prompt)
PROMPT_LEN = 1405


def _gen_cfg(cfg):
    from visrag_amd.evisrag import GenConfig
    return GenConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                     num_key_value_heads=cfg.num_key_value_heads, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
                     rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, mrope_section=tuple(cfg.mrope_section),
                     image_token_id=5, eos_token_ids=())


@pytest.fixture(scope="module")
def setup():
    from visrag_amd.evisrag import LLM, VisionConfig
    cfg = QwenGenConfig(num_hidden_layers=2)                       # every width is the 7B model's
    vcfg = QwenVisionConfig(depth=2, fullatt_block_indexes=(1,))   # block 0: 112-pixel windows, block 1: whole images
    wg = synth_weights(cfg, seed=0, device="cuda")                 # counter hash: the same numbers on the GPU and on the host
    wv = synth_vision_weights(vcfg, seed=1)
    vc = VisionConfig(depth=vcfg.depth, hidden_size=vcfg.hidden_size, num_heads=vcfg.num_heads, intermediate_size=vcfg.intermediate_size,
                      out_hidden_size=vcfg.out_hidden_size, window_size=vcfg.window_size,
                      fullatt_block_indexes=tuple(vcfg.fullatt_block_indexes))
    w = dict(wg)
    w.update(wv)
    llm = LLM(_gen_cfg(cfg), max_model_len=2048, max_prefill=1536, vision=vc, max_vision_rows=6144, weights=w)
    host = {k: v.cpu() for k, v in wg.items()}
    del w, wg
    torch.cuda.empty_cache()
    rng = np.random.default_rng(0)
    ids = rng.integers(16, cfg.vocab_size, PROMPT_LEN)
    n_img_tok = GRID[0] * GRID[1]
    at = 50
    for _ in range(N_IMG):
        ids[at:at + n_img_tok] = 5
        at += n_img_tok + 3                                        # three text tokens between pages
    embs = [(0.05 * rng.standard_normal((n_img_tok, cfg.hidden_size))).astype(np.float32) for _ in range(N_IMG)]
    yield cfg, vcfg, host, wv, llm, ids.tolist(), embs
    llm.close()


def _oracle_prefill(cfg, host, ids, embs, pos3):
    o = QwenGenOracle(cfg, host)
    idt = torch.tensor(ids)
    emb = o.embed(idt).clone()
    emb[idt == 5] = torch.from_numpy(np.concatenate(embs))
    return o, o.forward(emb, torch.from_numpy(pos3).long(), last_only=True)[-1]


def test_prefill_and_teacher_forced_decode_at_7b_shape(setup):
    """vg_prefill of the 1405-token five-page prompt (3-D positions from rope_index), then 16 teacher-forced vg_decode
    steps over the 1405+ row cache (16 KV ranges merged by their log-sum-exps; skinny GEMMs with the 7B K splits):
    logits within 1.5e-2 / 2e-2 of the logits' scale of the fp32 oracle, the engine's greedy pick the oracle's token
    or a near-tie in the oracle's penalised logits."""
    from visrag_amd.evisrag import SamplingParams
    cfg, vcfg, host, wv, llm, ids, embs = setup
    pos3 = llm.prefill(ids, embs, [GRID] * N_IMG)
    assert pos3.shape == (3, PROMPT_LEN) and int(pos3[1].max()) > int(pos3[0, 60])      # image rows carry h / w positions
    o, o_logits = _oracle_prefill(cfg, host, ids, embs, pos3)
    ours = llm.logits()
    ref = o_logits.numpy()
    scale = np.abs(ref).max()
    assert np.abs(ours - ref).max() < 1.5e-2 * scale, (np.abs(ours - ref).max(), scale)
    cos = float(ours @ ref / (np.linalg.norm(ours) * np.linalg.norm(ref)))
    assert cos > 1 - 2e-4, cos
    seen, nxt, exact = torch.tensor(ids), int(pos3.max()) + 1, 0
    sp = SamplingParams(temperature=0.0, repetition_penalty=1.05)
    for k in range(16):
        ours = torch.from_numpy(llm.logits())
        assert float((ours - o_logits).abs().max()) < 2e-2 * float(o_logits.abs().max()), k
        pick = llm.sample(sp, k)
        pen = apply_repetition_penalty(o_logits, seen, 1.05)
        want = int(torch.argmax(pen))
        if pick == want:
            exact += 1
        else:
            gap = float(pen[want] - pen[pick])
            assert 0 <= gap < 1e-2 * float(pen.abs().max()), (k, pick, want, gap)
        seen = torch.cat([seen, torch.tensor([want])])
        llm.decode(want, nxt)
        o_logits = o.forward(o.embed(torch.tensor([want])), torch.full((3, 1), nxt), last_only=True)[-1]
        nxt += 1
    assert exact >= 13, exact


def test_free_running_steps_equal_host_driven_steps_at_7b_shape(setup):
    """The captured decode + sample step (vg_run_step: hipGraph replay, state advancing on the device) against the
    one-call-pair-per-token loop, 24 tokens from the 1405-row cache, greedy and with temperature."""
    from visrag_amd.evisrag import SamplingParams
    cfg, vcfg, host, wv, llm, ids, embs = setup
    pr = [{"prompt_token_ids": ids, "multi_modal_data": {"image_embeds": embs, "image_grids": [GRID] * N_IMG}}]
    for temperature in (0.0, 0.7):
        sp = SamplingParams(temperature=temperature, repetition_penalty=1.05, max_tokens=24, seed=11, stop_token_ids=())
        a = llm.generate(pr, sp, pipelined=False)[0].outputs[0].token_ids
        b = llm.generate(pr, sp, pipelined=True)[0].outputs[0].token_ids
        assert len(a) == 24 and a == b, temperature


def test_decode_step_equals_prefill_of_longer_prompt_at_7b_shape(setup):
    cfg, vcfg, host, wv, llm, ids, embs = setup
    text = [t for t in ids if t != 5][:300]
    llm.prefill(text[:-1])
    llm.decode(text[-1], len(text) - 1)
    step = llm.logits()
    llm.prefill(text)
    full = llm.logits()
    assert np.abs(step - full).max() < 2e-2 * np.abs(full).max()


def test_sampler_over_152k_logits_vs_numpy(setup):
    """temperature 0: exactly numpy's argmax of the penalised logits (lowest id among ties); temperature > 0: the
    Gumbel-max pick is a sample of softmax(logits / T) — 4000 seeds against numpy's probabilities on the head of the
    distribution (two-stage reduction over 64 workgroups x 152064 logits)."""
    from visrag_amd.evisrag import SamplingParams
    cfg, vcfg, host, wv, llm, ids, embs = setup
    text = [t for t in ids if t != 5][:120]
    llm.prefill(text)
    logits = llm.logits().astype(np.float64)
    pen = apply_repetition_penalty(torch.from_numpy(logits), torch.tensor(text), 1.3).numpy()
    assert llm.sample(SamplingParams(temperature=0.0, repetition_penalty=1.3), 0) == int(np.argmax(pen))
    llm.prefill(text)                                              # (the pick above was marked seen)
    T, N = float(logits.std()) / 8.0, 4000                       # z = logits / T has a standard deviation of 8: a peaked head
    z = logits / T
    p = np.exp(z - z.max()); p /= p.sum()
    counts = np.zeros(cfg.vocab_size, dtype=np.int64)
    for seed in range(N):
        counts[llm.sample(SamplingParams(temperature=T, repetition_penalty=1.0, seed=seed), 0)] += 1
    head = np.argsort(p)[-12:]
    assert p[head].sum() > 0.2, p[head].sum()                      # the test has something to look at
    for i in head:
        sigma = np.sqrt(p[i] * (1 - p[i]) / N)
        assert abs(counts[i] / N - p[i]) < 4.5 * sigma + 2.0 / N, (int(i), counts[i] / N, p[i])
    assert abs(counts[head].sum() / N - p[head].sum()) < 4.5 * np.sqrt(p[head].sum() * (1 - p[head].sum()) / N)


def test_tower_at_7b_shape_on_a_5k_row_page(setup):
    """Two tower blocks at the 7B widths (1280 / 16 heads of 80 / 3420 -> 3584) on a 1008 x 1008 page (72 x 72 = 5184
    patch rows: 81 full 8 x 8-token windows, then one 5184-row full-attention segment) plus a small ragged page."""
    cfg, vcfg, host, wv, llm, ids, embs = setup
    grids = np.array([[1, 72, 72], [1, 28, 22]], dtype=np.int32)
    rows = int((grids[:, 1] * grids[:, 2]).sum())
    rng = np.random.default_rng(3)
    px = rng.standard_normal((rows, vcfg.patch_dim)).astype(np.float32)
    px = torch.from_numpy(px).to(torch.bfloat16).float().numpy()   # what the tower's bf16 input conversion sees exactly
    emb = llm.encode_images(px, grids)
    ref = QwenVisionOracle(vcfg, wv).forward(torch.from_numpy(px), [tuple(int(x) for x in g) for g in grids]).numpy()
    assert emb.shape == ref.shape == (rows // 4, cfg.hidden_size)
    scale = np.abs(ref).max()
    assert np.abs(emb - ref).max() < 2e-2 * scale, (np.abs(emb - ref).max(), scale)
    cos = (emb * ref).sum(-1) / (np.linalg.norm(emb, axis=-1) * np.linalg.norm(ref, axis=-1))
    assert cos.min() > 1 - 1e-3, cos.min()


def test_batched_decode_at_7b_shape(setup):
    """Five prompts (the five-page one and four text prompts of different lengths) decoded together on a five-slot model
    of the same weights: 12 tokens each, identical to one-at-a-time generation (M = 5 rows through the skinny GEMMs'
    7B K splits, per-sequence KV ranges at a 1405-row and at short caches in one attention launch)."""
    from visrag_amd.evisrag import LLM, SamplingParams
    cfg, vcfg, host, wv, llm, ids, embs = setup
    w = {k: v.cuda() for k, v in host.items()}
    many = LLM(_gen_cfg(cfg), max_model_len=2048, max_prefill=1536, weights=w, max_num_seqs=5)
    del w
    try:
        rng = np.random.default_rng(9)
        prompts = [{"prompt_token_ids": ids, "multi_modal_data": {"image_embeds": embs, "image_grids": [GRID] * N_IMG}}] + \
                  [{"prompt_token_ids": rng.integers(16, cfg.vocab_size, n).tolist()} for n in (40, 300, 129, 700)]
        sp = SamplingParams(temperature=0.0, repetition_penalty=1.05, max_tokens=12, stop_token_ids=())
        ref = [llm.generate([p], sp)[0].outputs[0].token_ids for p in prompts]
        got = [o.outputs[0].token_ids for o in many.generate(prompts, sp)]
        assert got == ref
    finally:
        many.close()
