"""-m gpu: the decode step's persistent layer kernel (csrc/gen_persist.hip: all decoder layers of a step in ONE launch,
phases separated by grid barriers, the next projection's weights prefetched under them) against the same step as separate
launches (VR_DECODE_PERSIST=0 when the model is finalized).  The kernel repeats the launches' arithmetic in their order, so
the logits must be EQUAL, bit for bit — at the fixtures' 256-wide shape (every K range one or two K-steps, one KV range)
and at the 7B widths over a 1405-row cache (14 / 14 / 18 K splits, 16 KV ranges, 28:4 grouped-query layout)."""
import os

import numpy as np
import pytest
import torch

from oracle.qwen_gen_oracle import QwenGenConfig, synth_weights

pytestmark = pytest.mark.gpu


def _llm(cfg, persist, **kw):
    from visrag_amd.evisrag import GenConfig, LLM
    g = GenConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                  num_key_value_heads=cfg.num_key_value_heads, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
                  rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, mrope_section=tuple(cfg.mrope_section),
                  image_token_id=5, eos_token_ids=())
    old = os.environ.get("VR_DECODE_PERSIST")
    os.environ["VR_DECODE_PERSIST"] = "1" if persist else "0"
    try:
        return LLM(g, weights=synth_weights(cfg, seed=3, device="cuda"), **kw)
    finally:
        if old is None:
            del os.environ["VR_DECODE_PERSIST"]
        else:
            os.environ["VR_DECODE_PERSIST"] = old


def _walk(llm, ids, steps, tokens):
    llm.prefill(ids)
    out = [llm.logits().copy()]
    nxt = len(ids)
    for k in range(steps):
        llm.decode(tokens[k], nxt)
        nxt += 1
        out.append(llm.logits().copy())
    return out


@pytest.mark.parametrize("shape", ["tiny", "7b"])
def test_persistent_decode_equals_separate_launches(shape):
    """Teacher-forced decode steps from prompts of several lengths (1 to 11 KV ranges; cache lengths on both sides of the
    range and tile boundaries), then free-running captured steps with temperature: equal logits, equal tokens.  (The two
    paths share their small arithmetic through csrc/gen_math.h: a rotation fused differently in the two kernels once showed
    up as one decode step in a few hundred with logits 2e-3 of their scale apart.)"""
    from visrag_amd.evisrag import SamplingParams
    if shape == "tiny":
        cfg = QwenGenConfig(hidden_size=256, num_hidden_layers=3, num_attention_heads=2, num_key_value_heads=1, intermediate_size=704,
                            vocab_size=1024, mrope_section=(16, 24, 24))
        kw = dict(max_model_len=512, max_prefill=256)
        lens, steps = [60, 127, 150, 255], 12
    else:
        cfg = QwenGenConfig(num_hidden_layers=2)
        kw = dict(max_model_len=2048, max_prefill=1536)
        lens, steps = [128, 383, 640, 1100, 1405], 8
    prompts = {}
    for n in lens:
        rng = np.random.default_rng(n)
        prompts[n] = (rng.integers(16, cfg.vocab_size, n).tolist(), rng.integers(16, cfg.vocab_size, steps).tolist())
    sp = SamplingParams(temperature=0.7, repetition_penalty=1.05, max_tokens=8, seed=11)
    n_run = lens[-1]

    def run(llm):
        walks = {n: _walk(llm, prompts[n][0], steps, prompts[n][1]) for n in lens}
        llm.prefill(prompts[n_run][0])
        first = llm.sample(sp, 0)
        llm.run_begin(n_run, sp)                     # the captured step (one graph launch per token)
        for _ in range(6):
            llm.run_step()
        toks = [first] + [llm.run_token(i) for i in range(6)]
        llm.run_end()
        return walks, toks

    a = _llm(cfg, False, **kw)
    ref, ref_run = run(a)
    a.close()
    del a
    torch.cuda.empty_cache()
    b = _llm(cfg, True, **kw)
    got, got_run = run(b)
    b.close()
    for n in lens:
        for k, (x, y) in enumerate(zip(ref[n], got[n])):
            assert np.isfinite(y).all(), (n, k)
            assert np.array_equal(x, y), (shape, n, k, float(np.abs(x - y).max()), float(np.abs(x).max()))
    assert got_run == ref_run
