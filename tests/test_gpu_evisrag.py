"""EVisRAG generator (language model) on the GPU against the HF-pinned oracle and fixtures:
prefill logits, greedy decoding with the repetition penalty, decode-vs-prefill consistency, the sampling kernel."""
import os

import numpy as np
import pytest
import torch

from oracle.qwen_gen_oracle import QwenGenOracle, apply_repetition_penalty, synth_weights, tiny_config

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "evisrag_tiny.npz")


def _gen_cfg(cfg):
    from visrag_amd.evisrag import GenConfig
    return GenConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                     num_key_value_heads=cfg.num_key_value_heads, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
                     rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, mrope_section=tuple(cfg.mrope_section),
                     image_token_id=5, eos_token_ids=())


@pytest.fixture(scope="module")
def setup():
    from visrag_amd.evisrag import LLM
    g = np.load(GOLD)
    cfg = tiny_config()
    w = synth_weights(cfg, seed=int(g["seed"]))
    llm = LLM(_gen_cfg(cfg), max_model_len=512, max_prefill=256, weights=w)
    yield g, cfg, w, llm
    llm.close()


def _prompt(g, tag):
    ids = g[f"{tag}_ids"].tolist()
    if tag == "b":
        return ids, [g["b_image_embeds"]], [(6, 4)]
    return ids, [], []


@pytest.mark.parametrize("tag", ["a", "b"])
def test_prefill_logits_match_reference(setup, tag):
    g, cfg, w, llm = setup
    ids, embs, grids = _prompt(g, tag)
    pos3 = llm.prefill(ids, embs, grids)
    np.testing.assert_array_equal(pos3, g[f"{tag}_pos3"])            # rope_index == the positions the fixture used
    ours, ref = llm.logits(), g[f"{tag}_logits"]
    # bf16 operands, fp32 accumulation and residual stream: a few 1e-3 of the logits' scale
    scale = np.abs(ref).max()
    assert np.abs(ours - ref).max() < 1.5e-2 * scale, (np.abs(ours - ref).max(), scale)
    cos = float(ours @ ref / (np.linalg.norm(ours) * np.linalg.norm(ref)))
    assert cos > 1 - 2e-4, cos


@pytest.mark.parametrize("tag", ["a", "b"])
def test_greedy_decoding_along_the_reference(setup, tag):
    """temperature 0, repetition_penalty 1.05, 24 tokens.  Teacher-forced along the HF run: at every step the
    engine's logits match the oracle's and its own pick is the reference token — or a near-tie in the oracle's
    penalised logits (bf16 operands against fp32 can flip those; random tiny weights make flat logits)."""
    from visrag_amd.evisrag import SamplingParams
    g, cfg, w, llm = setup
    ids, embs, grids = _prompt(g, tag)
    ref = g[f"{tag}_tokens"].tolist()
    o = QwenGenOracle(cfg, w)
    idt = torch.tensor(ids)
    emb = o.embed(idt).clone()
    if tag == "b":
        emb[8:32] = torch.from_numpy(g["b_image_embeds"])
    pos = torch.from_numpy(g[f"{tag}_pos3"]).long()
    o_logits = o.forward(emb, pos)[-1]
    pos3 = llm.prefill(ids, embs, grids)
    seen, nxt, exact = idt.clone(), int(pos3.max()) + 1, 0
    sp = SamplingParams(temperature=0.0, repetition_penalty=1.05)
    for k in range(24):
        ours = torch.from_numpy(llm.logits())
        assert float((ours - o_logits).abs().max()) < 2e-2 * float(o_logits.abs().max()), k
        pick = llm.sample(sp, k)
        pen = apply_repetition_penalty(o_logits, seen, 1.05)
        assert int(torch.argmax(pen)) == ref[k]                       # (the oracle reproduces HF token for token)
        if pick == ref[k]:
            exact += 1
        else:
            gap = float(pen[ref[k]] - pen[pick])
            assert 0 <= gap < 1e-2 * float(pen.abs().max()), (k, pick, ref[k], gap)
        seen = torch.cat([seen, torch.tensor([ref[k]])])
        llm.decode(ref[k], nxt)
        o_logits = o.forward(o.embed(torch.tensor([ref[k]])), torch.full((3, 1), nxt))[-1]
        nxt += 1
    assert exact >= 20, exact


def test_generate_call_site(setup):
    """llm.generate as predict.py:147 calls it: free-running greedy decode, prompt a reproduces the HF tokens."""
    from visrag_amd.evisrag import SamplingParams
    g, cfg, w, llm = setup
    out = llm.generate([{"prompt_token_ids": g["a_ids"].tolist()}],
                       SamplingParams(temperature=0.0, repetition_penalty=1.05, max_tokens=24))
    assert out[0].outputs[0].token_ids == g["a_tokens"].tolist()
    assert out[0].prompt_token_ids == g["a_ids"].tolist()


def test_decode_step_equals_prefill_of_longer_prompt(setup):
    """size-independent property: prefill(T) + one decode step == prefill(T + 1) on the last row's logits."""
    g, cfg, w, llm = setup
    ids = g["a_ids"].tolist()
    llm.prefill(ids[:-1])
    llm.decode(ids[-1], len(ids) - 1)
    step = llm.logits()
    llm.prefill(ids)
    full = llm.logits()
    assert np.abs(step - full).max() < 2e-2 * np.abs(full).max()
    assert int(step.argmax()) == int(full.argmax())


def test_decode_over_a_long_cache(setup):
    """a cache longer than 128 rows: the decode step's attention runs as several KV ranges merged by their
    log-sum-exps; against the oracle's full forward and against the engine's own prefill of the longer prompt"""
    g, cfg, w, llm = setup
    ids = np.random.default_rng(5).integers(0, cfg.vocab_size, 211).tolist()
    o = QwenGenOracle(cfg, w)
    idt = torch.tensor(ids)
    ref = o.forward(o.embed(idt), torch.arange(len(ids))[None].expand(3, -1))[-1].numpy()
    llm.prefill(ids[:-3])
    for k in range(3, 0, -1):
        llm.decode(ids[-k], len(ids) - k)
    step = llm.logits()
    llm.prefill(ids)
    full = llm.logits()
    scale = np.abs(ref).max()
    assert np.abs(step - ref).max() < 2e-2 * scale and np.abs(full - ref).max() < 2e-2 * scale
    assert np.abs(step - full).max() < 1e-2 * scale


def test_sampling_kernel(setup):
    from visrag_amd.evisrag import SamplingParams
    g, cfg, w, llm = setup
    ids = g["a_ids"].tolist()
    llm.prefill(ids)
    logits = torch.from_numpy(llm.logits())
    # temperature 0 == argmax of the penalised logits, exactly (same fp32 numbers)
    want = int(torch.argmax(apply_repetition_penalty(logits, torch.tensor(ids), 1.3)))
    assert llm.sample(SamplingParams(temperature=0.0, repetition_penalty=1.3), 0) == want
    # temperature > 0: reproducible per (seed, step), varies across seeds, stays on plausible tokens
    llm.prefill(ids)
    a = llm.sample(SamplingParams(temperature=1.0, repetition_penalty=1.0, seed=3), 0)
    llm.prefill(ids)
    b = llm.sample(SamplingParams(temperature=1.0, repetition_penalty=1.0, seed=3), 0)
    assert a == b
    picks = set()
    for seed in range(24):
        llm.prefill(ids)
        picks.add(llm.sample(SamplingParams(temperature=1.0, repetition_penalty=1.0, seed=seed), 0))
    assert len(picks) > 4
    top = set(torch.topk(logits, 200).indices.tolist())
    assert len(picks & top) >= len(picks) - 2          # softmax mass sits on the top logits


@pytest.mark.parametrize("temperature", [0.0, 0.7])
def test_free_running_steps_equal_host_driven_steps(setup, temperature):
    """generate(pipelined=True) — captured decode + sample steps replayed with token, position and cache length
    advancing on the device — produces the tokens of the one-call-pair-per-token loop, across the point where the
    decode attention starts cutting the cache into ranges (128 rows) and with sampling noise."""
    from visrag_amd.evisrag import SamplingParams
    g, cfg, w, llm = setup
    ids, embs, grids = _prompt(g, "b")
    sp = SamplingParams(temperature=temperature, repetition_penalty=1.05, max_tokens=140, seed=11, stop_token_ids=())
    pr = [{"prompt_token_ids": ids, "multi_modal_data": {"image_embeds": embs, "image_grids": grids}}]
    a = llm.generate(pr, sp, pipelined=False)[0].outputs[0].token_ids
    b = llm.generate(pr, sp, pipelined=True)[0].outputs[0].token_ids
    assert len(a) == 140 and a == b
    la = llm.logits()                                             # state after a free run is usable: logits of the last step
    assert np.isfinite(la).all()
    # a stop token ends the run early (one step may be in flight); the model is reusable afterwards
    stop = a[5]
    c = llm.generate(pr, SamplingParams(temperature=temperature, repetition_penalty=1.05, max_tokens=140, seed=11,
                                        stop_token_ids=(stop,)), pipelined=True)[0].outputs[0].token_ids
    assert c == a[:a.index(stop) + 1]
    assert llm.generate(pr, sp, pipelined=True)[0].outputs[0].token_ids == a


def test_run_api_errors(setup):
    from visrag_amd._lib import VisragHipError
    g, cfg, w, llm = setup
    ids, embs, grids = _prompt(g, "a")
    llm.prefill(ids, embs, grids)
    with pytest.raises(VisragHipError):
        llm.run_step()                                            # no run in progress
    from visrag_amd.evisrag import SamplingParams
    sp = SamplingParams(temperature=0.0, repetition_penalty=1.0)
    with pytest.raises(VisragHipError):
        llm.run_begin(len(ids), sp)                               # no sampled token on the device yet
    llm.sample(sp, 0)
    llm.run_begin(len(ids), sp)
    llm.run_step()
    with pytest.raises(VisragHipError):
        llm.run_token(1)                                          # only step 0 has been enqueued
    t0 = llm.run_token(0)
    assert 0 <= t0 < cfg.vocab_size
    llm.run_end()


def test_batched_generation_equals_one_at_a_time():
    """llm.generate with several prompts on a model that holds several sequences (max_num_seqs, vLLM's name): one
    vg_decode_batch step per token serves all of them — and every prompt gets exactly the tokens it gets alone (rows of
    a step are independent), for prompts of different lengths incl. one with an image block, a sequence that stops early,
    greedy and with temperature; more prompts than slots run in groups."""
    from visrag_amd.evisrag import LLM, SamplingParams
    g = np.load(GOLD)
    cfg = tiny_config()
    w = synth_weights(cfg, seed=int(g["seed"]))
    rng = np.random.default_rng(3)
    prompts = [{"prompt_token_ids": g["a_ids"].tolist()},
               {"prompt_token_ids": g["b_ids"].tolist(), "multi_modal_data": {"image_embeds": [g["b_image_embeds"]], "image_grids": [(6, 4)]}},
               {"prompt_token_ids": rng.integers(6, cfg.vocab_size, 150).tolist()},
               {"prompt_token_ids": rng.integers(6, cfg.vocab_size, 3).tolist()},
               {"prompt_token_ids": rng.integers(6, cfg.vocab_size, 77).tolist()}]
    one = LLM(_gen_cfg(cfg), max_model_len=512, max_prefill=256, weights=w)
    many = LLM(_gen_cfg(cfg), max_model_len=512, max_prefill=256, weights=w, max_num_seqs=3)
    try:
        for temperature in (0.0, 0.8):
            sp = SamplingParams(temperature=temperature, repetition_penalty=1.05, max_tokens=40, seed=5, stop_token_ids=())
            ref = [one.generate([p], sp)[0].outputs[0].token_ids for p in prompts]
            got = many.generate(prompts, sp)
            assert [o.outputs[0].token_ids for o in got] == ref, temperature
            assert [o.prompt_token_ids for o in got] == [p["prompt_token_ids"] for p in prompts]
            # a stop token ends one sequence early; the others run on
            stop = ref[2][7]
            sp2 = SamplingParams(temperature=temperature, repetition_penalty=1.05, max_tokens=40, seed=5, stop_token_ids=(stop,))
            ref2 = [one.generate([p], sp2)[0].outputs[0].token_ids for p in prompts[:3]]
            got2 = [o.outputs[0].token_ids for o in many.generate(prompts[:3], sp2)]
            assert got2 == ref2 and len(got2[2]) <= 8
        # the single-sequence entry points still work on a multi-slot model, on any slot
        many.select(2)
        many.prefill(g["a_ids"].tolist())
        one.prefill(g["a_ids"].tolist())
        np.testing.assert_array_equal(many.logits(), one.logits())
        many.select(0)
        with pytest.raises(Exception):
            many.select(3)
        with pytest.raises(Exception):
            many.decode_batch([0, 0], [7, 8], [5, 5])                     # a slot named twice
    finally:
        one.close(); many.close()
