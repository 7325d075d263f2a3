"""TEST INFRASTRUCTURE — CPU restatement (torch fp32) of the EVisRAG generator's language model: the Qwen2.5-VL text
decoder with multimodal RoPE, a KV cache, and the logits processing of the reference's sampling call.

Only tests/, __graft_entry__.smoke() and bench legs that time a CPU baseline may import this file.

The reference runs this arithmetic inside a third-party dependency that is NOT vendored:
`src/evisrag/predict.py:112-123,147` calls `vllm.LLM(model, dtype="bfloat16", limit_mm_per_prompt={"image": 5})`,
`SamplingParams(temperature, repetition_penalty=1.05, max_tokens=2048)` and `llm.generate` (vllm==0.9.1,
`EVisRAG_requirements.txt:236`; transformers==4.51.3 :222).  vLLM implements the published Qwen2.5-VL architecture;
this file restates it from the HuggingFace implementation installed in the build container
(transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py: RMSNorm :65-83, rotary tables :486-538, multimodal RoPE
:557-599, attention :602-690, decoder layer :692-758, SwiGLU MLP :541-555) and is pinned against that implementation
by the fixtures `oracle/gen_golden_evisrag.py` writes (tests/golden/evisrag_tiny.npz).  The reference repository has
no test or golden vector at this boundary: parity is pinned to HF, not to vLLM's outputs.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch


@dataclass
class QwenGenConfig:
    hidden_size: int = 3584
    num_hidden_layers: int = 28
    num_attention_heads: int = 28
    num_key_value_heads: int = 4
    intermediate_size: int = 18944
    vocab_size: int = 152064
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    mrope_section: Tuple[int, int, int] = (16, 24, 24)
    max_position_embeddings: int = 32768

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


def tiny_config() -> QwenGenConfig:
    """The fixture model: 2 layers, 2 query heads sharing 1 KV head of head_dim 128."""
    return QwenGenConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                         intermediate_size=512, vocab_size=1024, max_position_embeddings=512)


def _gen_config(cfg: QwenGenConfig):
    from visrag_amd.evisrag import GenConfig
    return GenConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                     num_key_value_heads=cfg.num_key_value_heads, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
                     rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, mrope_section=tuple(cfg.mrope_section))


def synth_weights(cfg: QwenGenConfig, seed: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    """The deterministic bf16-representable synthetic weights the product's tests and benchmarks load
    (visrag_amd.evisrag.gen_weight_specs / visrag_amd.synth's counter hash), identical on CPU and GPU."""
    from visrag_amd.evisrag import iter_synth_gen_weights
    return dict(iter_synth_gen_weights(_gen_config(cfg), seed, device=device))


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:   # modeling_qwen2_5_vl.py:65-83
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(v + eps))


def mrope_cos_sin(cfg: QwenGenConfig, pos3: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """pos3 [3][T] (temporal, height, width position of every token) -> cos, sin [T][head_dim] with the section
    selection of apply_multimodal_rotary_pos_emb already applied (:486-538, :589-594): channel pair p of the 64 takes
    the temporal position for p < 16, the height position for 16 <= p < 40, the width position beyond."""
    hd = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))        # [hd/2]
    sel = torch.repeat_interleave(torch.arange(3), torch.tensor(cfg.mrope_section))           # [hd/2] -> 0/1/2
    pos = pos3.float()[sel]                                                                   # [hd/2][T]
    fr = (pos * inv[:, None]).T                                                               # [T][hd/2]
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x: torch.Tensor) -> torch.Tensor:                                             # :153-157
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


class QwenGenOracle:
    """One sequence, KV cache as python lists; forward(embeds [T][E], pos3 [3][T]) -> logits of every new token."""

    def __init__(self, cfg: QwenGenConfig, weights: Dict[str, torch.Tensor]):
        self.cfg = cfg
        self.w = {k: v.float() for k, v in weights.items()}
        self.k_cache: List[Optional[torch.Tensor]] = [None] * cfg.num_hidden_layers
        self.v_cache: List[Optional[torch.Tensor]] = [None] * cfg.num_hidden_layers

    def reset(self):
        self.k_cache = [None] * self.cfg.num_hidden_layers
        self.v_cache = [None] * self.cfg.num_hidden_layers

    def embed(self, ids: torch.Tensor) -> torch.Tensor:
        return self.w["model.language_model.embed_tokens.weight"][ids.long()]

    def forward(self, x: torch.Tensor, pos3: torch.Tensor, last_only: bool = False) -> torch.Tensor:
        """last_only: logits of the last new token only ([1][vocab]) — a 1405-token prompt against a 152k vocabulary
        would otherwise spend most of its time and 0.9 GB on rows nobody reads."""
        c = self.cfg
        H, KV, hd = c.num_attention_heads, c.num_key_value_heads, c.head_dim
        T = x.shape[0]
        cos, sin = mrope_cos_sin(c, pos3)
        h = x.float()
        for l in range(c.num_hidden_layers):
            p = f"model.language_model.layers.{l}."
            w = self.w
            xn = rmsnorm(h, w[p + "input_layernorm.weight"], c.rms_norm_eps)
            q = (xn @ w[p + "self_attn.q_proj.weight"].T + w[p + "self_attn.q_proj.bias"]).view(T, H, hd)
            k = (xn @ w[p + "self_attn.k_proj.weight"].T + w[p + "self_attn.k_proj.bias"]).view(T, KV, hd)
            v = (xn @ w[p + "self_attn.v_proj.weight"].T + w[p + "self_attn.v_proj.bias"]).view(T, KV, hd)
            q = q * cos[:, None, :] + rotate_half(q) * sin[:, None, :]
            k = k * cos[:, None, :] + rotate_half(k) * sin[:, None, :]
            K = k if self.k_cache[l] is None else torch.cat([self.k_cache[l], k], 0)
            V = v if self.v_cache[l] is None else torch.cat([self.v_cache[l], v], 0)
            self.k_cache[l], self.v_cache[l] = K, V
            L = K.shape[0]
            g = H // KV
            # (query rows in blocks of 1024: the same per-row arithmetic, without a [heads][T][L] score tensor — 7.5 GB per
            # layer for an 8k-token prompt at 28 heads; prompts up to 1024 tokens, all the fixtures, take one block)
            Kr, Vr = K.repeat_interleave(g, dim=1), V.repeat_interleave(g, dim=1)
            a = torch.empty((T, H * hd), dtype=torch.float32)
            for t0 in range(0, T, 1024):
                t1 = min(T, t0 + 1024)
                s = torch.einsum("thd,lhd->htl", q[t0:t1], Kr) * hd ** -0.5                               # :186-208
                mask = torch.arange(L)[None, :] > (L - T + torch.arange(t0, t1))[:, None]                # causal
                s = s.masked_fill(mask[None], float("-inf"))
                a[t0:t1] = torch.einsum("htl,lhd->thd", torch.softmax(s, -1), Vr).reshape(t1 - t0, H * hd)
            h = h + a @ w[p + "self_attn.o_proj.weight"].T
            xn = rmsnorm(h, w[p + "post_attention_layernorm.weight"], c.rms_norm_eps)
            act = torch.nn.functional.silu(xn @ w[p + "mlp.gate_proj.weight"].T) * (xn @ w[p + "mlp.up_proj.weight"].T)
            h = h + act @ w[p + "mlp.down_proj.weight"].T
        hn = rmsnorm(h[-1:] if last_only else h, self.w["model.language_model.norm.weight"], c.rms_norm_eps)
        return hn @ self.w["lm_head.weight"].T


def apply_repetition_penalty(logits: torch.Tensor, seen: torch.Tensor, penalty: float) -> torch.Tensor:
    """vLLM / HF RepetitionPenaltyLogitsProcessor: for every token id that occurred in the prompt or the output so
    far, logit > 0 -> logit / penalty, else logit * penalty (predict.py:119-123: repetition_penalty=1.05)."""
    out = logits.clone()
    idx = torch.unique(seen.long())
    sel = out[idx]
    out[idx] = torch.where(sel > 0, sel / penalty, sel * penalty)
    return out


def greedy_generate(model: QwenGenOracle, prompt_ids: torch.Tensor, prompt_embeds: torch.Tensor, pos3: torch.Tensor,
                    max_new: int, penalty: float = 1.05, eos: Optional[int] = None) -> List[int]:
    """temperature 0 (predict.py's --temperature 0.0): argmax after the repetition penalty; new tokens continue the
    position counter at max(pos3) + 1 on all three axes (get_rope_index of the reference model family)."""
    model.reset()
    logits = model.forward(prompt_embeds, pos3)[-1]
    seen = prompt_ids.clone().long()
    nxt_pos = int(pos3.max()) + 1
    out: List[int] = []
    for _ in range(max_new):
        tok = int(torch.argmax(apply_repetition_penalty(logits, seen, penalty)))
        out.append(tok)
        if eos is not None and tok == eos:
            break
        seen = torch.cat([seen, torch.tensor([tok])])
        p = torch.full((3, 1), nxt_pos, dtype=torch.long)
        nxt_pos += 1
        logits = model.forward(model.embed(torch.tensor([tok])), p)[-1]
    return out
