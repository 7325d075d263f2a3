"""Generate tests/golden/evisrag_tiny.npz from the HuggingFace Qwen2.5-VL implementation installed in the BUILD
container (run here, never on the GPU box):

    python oracle/gen_golden_evisrag.py

Tiny language model (oracle/qwen_gen_oracle.tiny_config) with the deterministic synthetic weights, two prompts:
  A  40 text tokens, 1-D positions;
  B  8 text tokens + a 6 x 4 block of "image" embeddings (random vectors at placeholder ids, positions as
     get_rope_index lays an image out: temporal constant, height / width running over the grid) + 6 text tokens.
Stored: last-token prefill logits, and 24 greedy tokens with repetition_penalty 1.05 produced by stepping the HF
model with its KV cache and HF's own RepetitionPenaltyLogitsProcessor.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.qwen_gen_oracle import synth_weights, tiny_config  # noqa: E402


def build_hf(cfg, weights):
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    text = dict(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                num_key_value_heads=cfg.num_key_value_heads, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
                max_position_embeddings=cfg.max_position_embeddings, rms_norm_eps=cfg.rms_norm_eps, tie_word_embeddings=False,
                bos_token_id=None, eos_token_id=None,
                rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta, "mrope_section": list(cfg.mrope_section)})
    vision = dict(depth=1, hidden_size=64, intermediate_size=128, num_heads=2, out_hidden_size=cfg.hidden_size, patch_size=14,
                  spatial_merge_size=2, temporal_patch_size=2, window_size=56, fullatt_block_indexes=[0])
    m = Qwen2_5_VLForConditionalGeneration(Qwen2_5_VLConfig(text_config=text, vision_config=vision, bos_token_id=None,
                                                            eos_token_id=None)).eval().float()
    sd = m.state_dict()
    for k, v in weights.items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
        sd[k].copy_(v)
    return m


def hf_generate(m, ids, embeds, pos3, max_new, penalty):
    from transformers.generation.logits_process import RepetitionPenaltyLogitsProcessor
    proc = RepetitionPenaltyLogitsProcessor(penalty)
    with torch.no_grad():
        out = m(inputs_embeds=embeds[None], position_ids=pos3[:, None, :], use_cache=True)
        first_logits = out.logits[0, -1].clone()
        past, logits = out.past_key_values, out.logits[0, -1]
        seen = ids.clone()[None]
        nxt = int(pos3.max()) + 1
        toks = []
        for _ in range(max_new):
            tok = int(torch.argmax(proc(seen, logits[None])[0]))
            toks.append(tok)
            seen = torch.cat([seen, torch.tensor([[tok]])], 1)
            e = m.get_input_embeddings()(torch.tensor([[tok]]))
            out = m(inputs_embeds=e, position_ids=torch.full((3, 1, 1), nxt), past_key_values=past, use_cache=True)
            nxt += 1
            past, logits = out.past_key_values, out.logits[0, -1]
    return first_logits, toks


def main():
    cfg = tiny_config()
    w = synth_weights(cfg, seed=7)
    m = build_hf(cfg, w)
    g = torch.Generator().manual_seed(11)
    out = {}
    # prompt A: text only
    ids_a = torch.randint(0, cfg.vocab_size, (40,), generator=g)
    pos_a = torch.arange(40)[None].expand(3, 40).contiguous()
    emb_a = m.get_input_embeddings()(ids_a)
    # prompt B: text + image block + text
    ids_b = torch.cat([torch.randint(0, cfg.vocab_size, (8,), generator=g), torch.full((24,), 5),
                       torch.randint(0, cfg.vocab_size, (6,), generator=g)])
    emb_b = m.get_input_embeddings()(ids_b).clone()
    img = (torch.randn((24, cfg.hidden_size), generator=g) * 0.05).to(torch.bfloat16).float()
    emb_b[8:32] = img
    t = torch.cat([torch.arange(8), torch.full((24,), 8), torch.arange(6) + 8 + 6])          # text resumes at max + 1
    hh = torch.cat([torch.arange(8), 8 + torch.arange(6).repeat_interleave(4), torch.arange(6) + 14])
    ww = torch.cat([torch.arange(8), 8 + torch.arange(4).repeat(6), torch.arange(6) + 14])
    pos_b = torch.stack([t, hh, ww])
    for tag, ids, emb, pos in (("a", ids_a, emb_a, pos_a), ("b", ids_b, emb_b, pos_b)):
        logits, toks = hf_generate(m, ids, emb.detach(), pos, 24, 1.05)
        out[f"{tag}_ids"] = ids.numpy().astype(np.int32)
        out[f"{tag}_pos3"] = pos.numpy().astype(np.int32)
        out[f"{tag}_logits"] = logits.numpy().astype(np.float32)
        out[f"{tag}_tokens"] = np.array(toks, dtype=np.int32)
    out["b_image_embeds"] = img.numpy().astype(np.float32)
    out["seed"] = np.array(7)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "evisrag_tiny.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
