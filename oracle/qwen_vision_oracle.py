"""TEST INFRASTRUCTURE — CPU restatement (torch fp32) of the EVisRAG generator's VISION TOWER: the Qwen2.5-VL window
attention ViT that turns page images into the embedding rows the language model's prefill consumes
(include/visrag_gen.h: vg_prefill's `embeds`).

Only tests/, __graft_entry__.smoke() and bench legs that time a CPU baseline may import this file.

Where it sits in the reference: `src/evisrag/predict.py:98-103` puts up to five page images in a chat message,
`:140` `process_vision_info(msg)` loads them, `:147` `llm.generate` hands them to vLLM (vllm==0.9.1, not vendored),
whose Qwen2.5-VL model runs the image processor (smart_resize, normalise, patchify) and this tower.  The arithmetic is
restated from the HuggingFace implementation installed in the build container
(transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py: patch embed :99-126, rotary table :129-141, merger :144-158,
rotate-half on q and k :161-175, attention over packed windows :211-283, block :286-325, tower forward :406-470;
transformers/vision_utils.py: position ids :81-127, window order :130-188;
transformers/models/qwen2_vl/image_processing_qwen2_vl.py: smart_resize :62-88, patchify :164-198)
and pinned to it by the fixtures `oracle/gen_golden_evisrag_vision.py` writes (tests/golden/evisrag_vision_tiny.npz).
The reference repository has no test or golden vector at this boundary: parity is pinned to HF, not to vLLM's outputs.

Status: oracle only.  The HIP vision tower is the next piece of SURVEY.md section 8f row 4; this file and its fixture
are the checker it will be built against.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Iterator, List, Sequence, Tuple

import numpy as np
import torch


@dataclass
class QwenVisionConfig:
    """Qwen2.5-VL-7B's tower by default."""
    depth: int = 32
    hidden_size: int = 1280
    num_heads: int = 16
    intermediate_size: int = 3420
    out_hidden_size: int = 3584
    in_channels: int = 3
    patch_size: int = 14
    temporal_patch_size: int = 2
    spatial_merge_size: int = 2
    window_size: int = 112
    fullatt_block_indexes: Tuple[int, ...] = (7, 15, 23, 31)
    rms_norm_eps: float = 1e-6

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def patch_dim(self) -> int:
        return self.in_channels * self.temporal_patch_size * self.patch_size * self.patch_size

    @property
    def merge_unit(self) -> int:
        return self.spatial_merge_size ** 2


def tiny_vision_config(out_hidden_size: int = 256) -> QwenVisionConfig:
    """The fixture tower: 3 blocks (block 1 attends over whole images, 0 and 2 over 2 x 2-token windows), 2 heads of
    head_dim 40 — like the 7B tower's 80 not a power of two — and an odd SwiGLU width."""
    return QwenVisionConfig(depth=3, hidden_size=80, num_heads=2, intermediate_size=108, out_hidden_size=out_hidden_size,
                            window_size=56, fullatt_block_indexes=(1,))


def hd80_vision_config(out_hidden_size: int = 256) -> QwenVisionConfig:
    """A second fixture tower with the 7B tower's head_dim (80): 2 blocks, the second one attending over whole images."""
    return QwenVisionConfig(depth=2, hidden_size=160, num_heads=2, intermediate_size=216, out_hidden_size=out_hidden_size,
                            window_size=56, fullatt_block_indexes=(1,))


PREFIX = "model.visual."


def vision_weight_specs(cfg: QwenVisionConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(state-dict key, shape, kind) of every tensor of the tower, HF names.  kind: "w" matrix, "b" bias, "g" norm gain."""
    H, I, P = cfg.hidden_size, cfg.intermediate_size, cfg.patch_size
    out = [(PREFIX + "patch_embed.proj.weight", (H, cfg.in_channels, cfg.temporal_patch_size, P, P), "w")]
    for i in range(cfg.depth):
        b = f"{PREFIX}blocks.{i}."
        out += [(b + "norm1.weight", (H,), "g"), (b + "norm2.weight", (H,), "g"),
                (b + "attn.qkv.weight", (3 * H, H), "w"), (b + "attn.qkv.bias", (3 * H,), "b"),
                (b + "attn.proj.weight", (H, H), "w"), (b + "attn.proj.bias", (H,), "b"),
                (b + "mlp.gate_proj.weight", (I, H), "w"), (b + "mlp.gate_proj.bias", (I,), "b"),
                (b + "mlp.up_proj.weight", (I, H), "w"), (b + "mlp.up_proj.bias", (I,), "b"),
                (b + "mlp.down_proj.weight", (H, I), "w"), (b + "mlp.down_proj.bias", (H,), "b")]
    M = H * cfg.merge_unit
    out += [(PREFIX + "merger.ln_q.weight", (H,), "g"),
            (PREFIX + "merger.mlp.0.weight", (M, M), "w"), (PREFIX + "merger.mlp.0.bias", (M,), "b"),
            (PREFIX + "merger.mlp.2.weight", (cfg.out_hidden_size, M), "w"), (PREFIX + "merger.mlp.2.bias", (cfg.out_hidden_size,), "b")]
    return out


def synth_vision_weights(cfg: QwenVisionConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Deterministic bf16-representable weights (torch's CPU generator: this is a fixture recipe, not the product's
    counter hash — the tensors themselves are stored in the fixture)."""
    g = torch.Generator().manual_seed(seed)
    w = {}
    for name, shape, kind in vision_weight_specs(cfg):
        if kind == "w":
            fan_in = int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
        elif kind == "b":
            t = torch.randn(shape, generator=g) * 0.05
        else:
            t = 1.0 + torch.randn(shape, generator=g) * 0.1
        w[name] = t.to(torch.bfloat16).float()
    return w


# ---------------------------------------------------------------------------------------------------------------
# image side: smart_resize, patchify
# ---------------------------------------------------------------------------------------------------------------

def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56,
                 max_pixels: int = 14 * 14 * 4 * 1280) -> Tuple[int, int]:
    """Target size of the processor's resize (image_processing_qwen2_vl.py:62-88): both sides multiples of `factor`
    (= patch * merge), the area inside [min_pixels, max_pixels], aspect ratio kept as well as the grid allows."""
    if max(height, width) / min(height, width) > 200:
        raise ValueError("absolute aspect ratio must be smaller than 200")
    hb, wb = round(height / factor) * factor, round(width / factor) * factor
    if hb * wb > max_pixels:
        beta = math.sqrt(height * width / max_pixels)
        hb = max(factor, math.floor(height / beta / factor) * factor)
        wb = max(factor, math.floor(width / beta / factor) * factor)
    elif hb * wb < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        hb = math.ceil(height * beta / factor) * factor
        wb = math.ceil(width * beta / factor) * factor
    return hb, wb


def patchify(image: torch.Tensor, cfg: QwenVisionConfig) -> Tuple[torch.Tensor, Tuple[int, int, int]]:
    """image [C][H][W] (already resized and normalised) -> pixel rows [gh*gw][C*tp*p*p] and its grid (1, gh, gw)
    (image_processing_qwen2_vl.py:164-198).  A still image fills every temporal slot of a patch with the same pixels.
    Row order is merge-block major: (gh/m, gw/m, m, m); feature order (C, tp, p, p)."""
    C, H, W = image.shape
    p, m, tp = cfg.patch_size, cfg.spatial_merge_size, cfg.temporal_patch_size
    gh, gw = H // p, W // p
    x = image.reshape(C, gh // m, m, p, gw // m, m, p).permute(1, 4, 2, 5, 0, 3, 6)      # [gh/m][gw/m][m][m][C][p][p]
    x = x.unsqueeze(5).expand(-1, -1, -1, -1, -1, tp, -1, -1)
    return x.reshape(gh * gw, C * tp * p * p).contiguous(), (1, gh, gw)


# ---------------------------------------------------------------------------------------------------------------
# token geometry: rotary positions, window order, attention segments
# ---------------------------------------------------------------------------------------------------------------

def position_hw(grids: Sequence[Tuple[int, int, int]], m: int) -> torch.Tensor:
    """[rows][2] (h, w) patch coordinate of every pixel row, in patchify's merge-block-major order (vision_utils.py:81-127)."""
    out = []
    for t, gh, gw in grids:
        hh = torch.arange(gh)[:, None].expand(gh, gw).reshape(gh // m, m, gw // m, m).permute(0, 2, 1, 3).reshape(-1)
        ww = torch.arange(gw)[None, :].expand(gh, gw).reshape(gh // m, m, gw // m, m).permute(0, 2, 1, 3).reshape(-1)
        out.append(torch.stack([hh, ww], -1).repeat(t, 1))
    return torch.cat(out)


def window_order(grids: Sequence[Tuple[int, int, int]], cfg: QwenVisionConfig) -> Tuple[torch.Tensor, List[int]]:
    """The permutation of merged tokens (groups of m*m rows) that makes every attention window contiguous, and the
    row boundaries of the windows (vision_utils.py:130-188).  A window is `window_size / patch / m` merged tokens on a
    side; windows on the right / bottom edge are smaller.  When a side is an exact multiple of the window, HF still
    pads by a full window — those windows are empty and vanish from the boundaries."""
    m = cfg.spatial_merge_size
    ws = cfg.window_size // m // cfg.patch_size
    order, bounds, base = [], [0], 0
    for t, gh, gw in grids:
        lh, lw = gh // m, gw // m
        idx = torch.arange(t * lh * lw).reshape(t, lh, lw)
        nh, nw = lh // ws + 1, lw // ws + 1
        pad = torch.full((t, nh * ws, nw * ws), -1, dtype=torch.long)
        pad[:, :lh, :lw] = idx
        win = pad.reshape(t, nh, ws, nw, ws).permute(0, 1, 3, 2, 4).reshape(t * nh * nw, ws * ws)
        for wrow in win:
            keep = wrow[wrow >= 0]
            if keep.numel():
                order.append(keep + base)
                bounds.append(bounds[-1] + keep.numel() * m * m)
        base += t * lh * lw
    return torch.cat(order), bounds


def image_bounds(grids: Sequence[Tuple[int, int, int]]) -> List[int]:
    """Row boundaries of full attention: one segment per frame (vision_utils.py:42-65)."""
    b = [0]
    for t, gh, gw in grids:
        for _ in range(t):
            b.append(b[-1] + gh * gw)
    return b


# ---------------------------------------------------------------------------------------------------------------
# the tower
# ---------------------------------------------------------------------------------------------------------------

def _rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def _rot_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], -1)


class QwenVisionOracle:
    def __init__(self, cfg: QwenVisionConfig, weights: Dict[str, torch.Tensor]):
        self.cfg = cfg
        self.w = {k: v.float() for k, v in weights.items() if k.startswith(PREFIX)}
        missing = [n for n, _, _ in vision_weight_specs(cfg) if n not in self.w]
        if missing:
            raise KeyError(f"vision weights missing: {missing[:4]}")

    def rotary(self, grids) -> Tuple[torch.Tensor, torch.Tensor]:
        """cos, sin [rows][head_dim]: the first quarter of the channel PAIRS turns with the row's h coordinate, the second
        quarter with w, and rotate-half pairs channel c with c + head_dim/2 (:129-141, :161-175, :441-446)."""
        hd = self.cfg.head_dim
        inv = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float32) / (hd // 2)))      # [hd/4]
        pos = position_hw(grids, self.cfg.spatial_merge_size).float()                                # [rows][2]
        fr = (pos[:, :, None] * inv[None, None, :]).reshape(pos.shape[0], -1)                        # [rows][hd/2]: h block, w block
        emb = torch.cat([fr, fr], -1)
        return emb.cos(), emb.sin()

    def forward(self, pixels: torch.Tensor, grids: Sequence[Tuple[int, int, int]], return_rows: bool = False):
        """pixels [rows][C*tp*p*p] (patchify's order, images concatenated) -> [rows / m^2][out_hidden] embedding rows in
        the images' own token order (what replaces the <|image_pad|> placeholders)."""
        cfg, w = self.cfg, self.w
        m2, H, nh, hd = cfg.merge_unit, cfg.hidden_size, cfg.num_heads, cfg.head_dim
        rows = pixels.shape[0]
        # a Conv3d whose kernel equals its stride is a matrix product over the flattened patch (:120-126)
        x = pixels.float() @ w[PREFIX + "patch_embed.proj.weight"].reshape(H, -1).T
        order, wbounds = window_order(grids, cfg)
        fbounds = image_bounds(grids)
        perm = (order[:, None] * m2 + torch.arange(m2)[None, :]).reshape(-1)         # rows follow their merged token
        x = x[perm]
        cos, sin = self.rotary(grids)
        cos, sin = cos[perm], sin[perm]
        for i in range(cfg.depth):
            b = f"{PREFIX}blocks.{i}."
            bounds = fbounds if i in cfg.fullatt_block_indexes else wbounds
            h = _rmsnorm(x, w[b + "norm1.weight"], cfg.rms_norm_eps)
            qkv = (h @ w[b + "attn.qkv.weight"].T + w[b + "attn.qkv.bias"]).reshape(rows, 3, nh, hd)
            q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
            q = q * cos[:, None, :] + _rot_half(q) * sin[:, None, :]
            k = k * cos[:, None, :] + _rot_half(k) * sin[:, None, :]
            att = torch.empty_like(q)
            for s, e in zip(bounds[:-1], bounds[1:]):                                # bidirectional inside a segment
                sc = torch.einsum("qhd,khd->hqk", q[s:e], k[s:e]) / math.sqrt(hd)
                att[s:e] = torch.einsum("hqk,khd->qhd", torch.softmax(sc, -1), v[s:e])
            x = x + att.reshape(rows, H) @ w[b + "attn.proj.weight"].T + w[b + "attn.proj.bias"]
            h = _rmsnorm(x, w[b + "norm2.weight"], cfg.rms_norm_eps)
            gate = h @ w[b + "mlp.gate_proj.weight"].T + w[b + "mlp.gate_proj.bias"]
            up = h @ w[b + "mlp.up_proj.weight"].T + w[b + "mlp.up_proj.bias"]
            x = x + (torch.nn.functional.silu(gate) * up) @ w[b + "mlp.down_proj.weight"].T + w[b + "mlp.down_proj.bias"]
        # merger (:144-158): RMSNorm per row, then the m*m rows of a merged token side by side through a GELU MLP
        h = _rmsnorm(x, w[PREFIX + "merger.ln_q.weight"], cfg.rms_norm_eps).reshape(rows // m2, m2 * H)
        h = torch.nn.functional.gelu(h @ w[PREFIX + "merger.mlp.0.weight"].T + w[PREFIX + "merger.mlp.0.bias"])
        out = h @ w[PREFIX + "merger.mlp.2.weight"].T + w[PREFIX + "merger.mlp.2.bias"]
        out = out[torch.argsort(order)]                                              # back to image order
        if return_rows:
            inv_perm = torch.argsort(perm)
            return out, x[inv_perm]
        return out
