"""The generator oracle (oracle/qwen_gen_oracle.py) against the fixtures the HuggingFace Qwen2.5-VL implementation
produced in the build container (oracle/gen_golden_evisrag.py -> tests/golden/evisrag_tiny.npz)."""
import os

import numpy as np
import torch

from oracle.qwen_gen_oracle import (QwenGenOracle, apply_repetition_penalty, greedy_generate, mrope_cos_sin, synth_weights,
                                    tiny_config)

GOLD = os.path.join(os.path.dirname(__file__), "golden", "evisrag_tiny.npz")


def _prompt(g, o, tag):
    ids = torch.from_numpy(g[f"{tag}_ids"]).long()
    emb = o.embed(ids).clone()
    if tag == "b":
        emb[8:32] = torch.from_numpy(g["b_image_embeds"])
    return ids, emb, torch.from_numpy(g[f"{tag}_pos3"]).long()


def test_oracle_prefill_logits_and_greedy_tokens_match_hf():
    g = np.load(GOLD)
    cfg = tiny_config()
    o = QwenGenOracle(cfg, synth_weights(cfg, seed=int(g["seed"])))
    for tag in ("a", "b"):
        ids, emb, pos = _prompt(g, o, tag)
        o.reset()
        logits = o.forward(emb, pos)[-1]
        np.testing.assert_allclose(logits.numpy(), g[f"{tag}_logits"], rtol=1e-4, atol=2e-5)
        toks = greedy_generate(o, ids, emb, pos, max_new=24, penalty=1.05)
        assert toks == g[f"{tag}_tokens"].tolist()


def test_mrope_sections_and_text_positions():
    cfg = tiny_config()
    pos = torch.tensor([[3, 7], [3, 9], [3, 11]])
    cos, sin = mrope_cos_sin(cfg, pos)
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, 128, 2).float() / 128))
    # token 0: all three axes equal -> plain 1-D RoPE
    np.testing.assert_allclose(cos[0, :64].numpy(), torch.cos(3 * inv).numpy(), rtol=1e-6, atol=1e-7)
    # token 1: pair p takes t (p < 16), h (16 <= p < 40), w (p >= 40); both halves of the head carry the same angle
    exp = torch.cat([7 * inv[:16], 9 * inv[16:40], 11 * inv[40:]])
    np.testing.assert_allclose(sin[1, :64].numpy(), torch.sin(exp).numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(sin[1, 64:].numpy(), sin[1, :64].numpy())


def test_repetition_penalty_rule():
    l = torch.tensor([2.0, -2.0, 1.0, -1.0])
    out = apply_repetition_penalty(l, torch.tensor([0, 1, 1]), 1.05)
    np.testing.assert_allclose(out.numpy(), [2.0 / 1.05, -2.0 * 1.05, 1.0, -1.0], rtol=1e-6)
