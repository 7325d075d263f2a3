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
    from visrag_amd.evisrag import SamplingParams
    if shape == "tiny":
        cfg = QwenGenConfig(hidden_size=256, num_hidden_layers=3, num_attention_heads=2, num_key_value_heads=1, intermediate_size=704,
                            vocab_size=1024, mrope_section=(16, 24, 24))
        kw = dict(max_model_len=512, max_prefill=256)
        n_prompt, steps = 150, 12
    else:
        cfg = QwenGenConfig(num_hidden_layers=2)
        kw = dict(max_model_len=2048, max_prefill=1536)
        n_prompt, steps = 1405, 10
    rng = np.random.default_rng(1)
    ids = rng.integers(16, cfg.vocab_size, n_prompt).tolist()
    toks = rng.integers(16, cfg.vocab_size, steps).tolist()
    a = _llm(cfg, False, **kw)
    ref = _walk(a, ids, steps, toks)
    # free-running steps of the separate-launch model, for the captured-graph comparison below
    sp = SamplingParams(temperature=0.7, repetition_penalty=1.05, max_tokens=8, seed=11)
    a.prefill(ids)
    first = a.sample(sp, 0)
    a.run_begin(n_prompt, sp)
    for _ in range(6):
        a.run_step()
    ref_run = [a.run_token(i) for i in range(6)]
    a.run_end()
    a.close()
    del a
    torch.cuda.empty_cache()
    b = _llm(cfg, True, **kw)
    got = _walk(b, ids, steps, toks)
    for k, (x, y) in enumerate(zip(ref, got)):
        assert np.isfinite(y).all(), k
        assert np.array_equal(x, y), (shape, k, float(np.abs(x - y).max()), float(np.abs(x).max()))
    # the captured step (one graph launch per token) runs the same kernel
    b.prefill(ids)
    assert b.sample(sp, 0) == first
    b.run_begin(n_prompt, sp)
    for _ in range(6):
        b.run_step()
    assert [b.run_token(i) for i in range(6)] == ref_run
    b.run_end()
    b.close()
