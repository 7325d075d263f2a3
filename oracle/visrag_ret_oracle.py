"""TEST INFRASTRUCTURE ONLY — CPU oracle for the VisRAG-Ret embedding + retrieval hot path.

A plain fp32 (torch-on-CPU / numpy) restatement of the reference's algorithm, one function
per reference function, each citing the reference file:line it follows (paths relative to
/root/reference).  It exists so that the HIP path can be checked on the GPU box, where
/root/reference does not exist.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import it; the product package `visrag_amd/` never does.

Pinning: the reference has NO golden vectors for this path, so the
oracle is pinned against outputs of the reference's own code run in the build container
(oracle/ref_harness.py + oracle/gen_golden.py -> tests/golden/*.npz, checked by
tests/test_oracle_golden.py).  Third-party arithmetic the reference relies on and that is
restated here from its published definition: torch SDPA (softmax(QK^T/sqrt(d))V),
nn.MultiheadAttention, F.interpolate(bicubic, antialias=True) [called, not restated],
HF `_prepare_4d_causal_attention_mask_for_sdpa` (causal AND key-padding mask).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- host ---
def ensure_divide(length, patch_size):
    # modeling_minicpmv.py:540-541
    return max(round(length / patch_size) * patch_size, patch_size)


def find_best_resize(original_size, scale_resolution, patch_size, allow_upscale=False):
    # modeling_minicpmv.py:544-552
    width, height = original_size
    if (width * height > scale_resolution * scale_resolution) or allow_upscale:
        r = width / height
        height = int(scale_resolution / math.sqrt(r))
        width = int(height * r)
    return (ensure_divide(width, patch_size), ensure_divide(height, patch_size))


def get_refine_size(original_size, grid, scale_resolution, patch_size, allow_upscale=False):
    # modeling_minicpmv.py:555-576
    width, height = original_size
    grid_x, grid_y = grid
    refine_width = ensure_divide(width, grid_x)
    refine_height = ensure_divide(height, grid_y)
    best = find_best_resize((refine_width / grid_x, refine_height / grid_y),
                            scale_resolution, patch_size, allow_upscale=allow_upscale)
    return (best[0] * grid_x, best[1] * grid_y)


def slice_plan(size, max_slice_nums=9, scale_resolution=448, patch_size=14):
    """Sizes only (no pixels): returns (source_size, refine_size|None, best_grid|None).
    modeling_minicpmv.py:482-537."""
    w, h = size
    log_ratio = math.log(w / h)
    ratio = w * h / (scale_resolution * scale_resolution)
    multiple = min(math.ceil(ratio), max_slice_nums)
    if multiple <= 1:
        return find_best_resize(size, scale_resolution, patch_size, allow_upscale=True), None, None
    cands = [i for i in (multiple - 1, multiple, multiple + 1) if i != 1 and i <= max_slice_nums]
    source = find_best_resize(size, scale_resolution, patch_size)
    grids = []
    for n in cands:
        for m in range(1, n + 1):
            if n % m == 0:
                grids.append([m, n // m])
    best_grid, min_err = [1, 1], float("inf")
    for g in grids:
        err = abs(log_ratio - math.log(g[0] / g[1]))
        if err < min_err:
            best_grid, min_err = g, err
    refine = get_refine_size(size, best_grid, scale_resolution, patch_size, allow_upscale=True)
    return source, refine, best_grid


def to_pixel_tensor(u8_hwc: np.ndarray) -> torch.Tensor:
    """ToTensor + Normalize(0.5, 0.5): modeling_minicpmv.py:84-92
    (IMAGENET_INCEPTION_MEAN/STD = 0.5).  u8 [H,W,3] -> f32 [3,H,W]."""
    t = torch.from_numpy(np.array(u8_hwc, copy=True)).permute(2, 0, 1).to(torch.float32).div(255)
    return (t - 0.5) / 0.5


# ------------------------------------------------------------------------------ ViT ---
def resample_abs_pos_embed(posemb: torch.Tensor, new_hw: Tuple[int, int]) -> torch.Tensor:
    """timm/layers/pos_embed.py:17-57 (num_prefix_tokens=0, bicubic, antialias=True).
    posemb [1, G*G, D] -> [1, h*w, D]."""
    n = posemb.shape[1]
    g = int(math.sqrt(n))
    if new_hw[0] * new_hw[1] == n and new_hw[0] == new_hw[1]:
        return posemb
    D = posemb.shape[-1]
    p = posemb.float().reshape(1, g, g, D).permute(0, 3, 1, 2)
    p = F.interpolate(p, size=tuple(new_hw), mode="bicubic", antialias=True)
    return p.permute(0, 2, 3, 1).reshape(1, -1, D)


def layer_norm(x, w, b, eps):
    # nn.LayerNorm(eps=1e-6): vision_transformer.py:465 ; resampler.py:111
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def sdpa(q, k, v, mask=None, scale=None):
    """softmax(q k^T * scale + mask) v, fp32; the definition of
    F.scaled_dot_product_attention used at vision_transformer.py:92-96 and
    modeling_minicpm.py:895-903."""
    scale = scale if scale is not None else q.shape[-1] ** -0.5
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    if mask is not None:
        s = s + mask
    return torch.matmul(torch.softmax(s, dim=-1), v)


def vit_forward(W: Dict[str, torch.Tensor], cfg, pixels: torch.Tensor, taps=None) -> torch.Tensor:
    """VisionTransformer.forward_features minus the dropped last block:
    vision_transformer.py:682-692 (patch_embed -> _pos_embed -> blocks -> norm),
    patch_embed.py:68-93 (Conv2d k=s=14 with bias, NHWC out), _pos_embed :600-631,
    Block :165-168, Attention :86-107, Mlp (erf GELU) mlp.py:41-47.
    pixels [B,3,H,W] f32 -> [B, h*w, D]."""
    P, D, H = cfg.patch_size, cfg.vit_dim, cfg.vit_heads
    B = pixels.shape[0]
    x = F.conv2d(pixels, W["vpm.patch_embed.proj.weight"], W["vpm.patch_embed.proj.bias"], stride=P)
    gh, gw = x.shape[-2:]
    x = x.permute(0, 2, 3, 1).reshape(B, gh * gw, D)
    x = x + resample_abs_pos_embed(W["vpm.pos_embed"], (gh, gw))
    if taps is not None:
        taps["vit_embed"] = x.clone()
    hd = D // H
    for n in range(cfg.vit_depth):
        p = f"vpm.blocks.{n}."
        y = layer_norm(x, W[p + "norm1.weight"], W[p + "norm1.bias"], cfg.vit_ln_eps)
        qkv = F.linear(y, W[p + "attn.qkv.weight"], W[p + "attn.qkv.bias"])
        qkv = qkv.reshape(B, -1, 3, H, hd).permute(2, 0, 3, 1, 4)
        a = sdpa(qkv[0], qkv[1], qkv[2])
        a = a.transpose(1, 2).reshape(B, -1, D)
        x = x + F.linear(a, W[p + "attn.proj.weight"], W[p + "attn.proj.bias"])
        y = layer_norm(x, W[p + "norm2.weight"], W[p + "norm2.bias"], cfg.vit_ln_eps)
        y = F.gelu(F.linear(y, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"]))  # exact erf
        x = x + F.linear(y, W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"])
        if taps is not None and n == 0:
            taps["vit_block0"] = x.clone()
    x = layer_norm(x, W["vpm.norm.weight"], W["vpm.norm.bias"], cfg.vit_ln_eps)
    if taps is not None:
        taps["vit_out"] = x.clone()
    return x


# ------------------------------------------------------------------------- resampler ---
def sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    # resampler.py:71-90
    omega = np.arange(embed_dim // 2, dtype=np.float32)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(embed_dim: int, grid_hw) -> np.ndarray:
    """resampler.py:38-68: grid = meshgrid(w, h) (w first); first half of the channels
    encodes grid[0] (= the column index), second half grid[1] (= the row index)."""
    gh, gw = (grid_hw, grid_hw) if isinstance(grid_hw, int) else grid_hw
    grid_h = np.arange(gh, dtype=np.float32)
    grid_w = np.arange(gw, dtype=np.float32)
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape(2, 1, gh, gw)
    emb_h = sincos_1d(embed_dim // 2, grid[0])
    emb_w = sincos_1d(embed_dim // 2, grid[1])
    return np.concatenate([emb_h, emb_w], axis=1)


def resampler_forward(W, cfg, x: torch.Tensor, tgt_hw, taps=None) -> torch.Tensor:
    """Resampler.forward (adaptive=True): resampler.py:146-168; nn.MultiheadAttention with
    q = ln_q(query)+pos8x8, k = ln_kv(kv_proj(x))+pos(tgt), v = ln_kv(kv_proj(x)).
    x [B,N,Dv] -> [B,Q,E]."""
    E, nh = cfg.hidden_size, cfg.resampler_heads
    hd = E // nh
    B, N, _ = x.shape
    eps = cfg.resampler_ln_eps
    pos_kv = torch.from_numpy(sincos_2d(E, tgt_hw)).float()                  # [N,E]
    g = int(math.sqrt(cfg.query_num))
    pos_q = torch.from_numpy(sincos_2d(E, g)).float()                        # [Q,E]
    x = F.linear(x, W["resampler.kv_proj.weight"])
    x = layer_norm(x, W["resampler.ln_kv.weight"], W["resampler.ln_kv.bias"], eps)
    q = layer_norm(W["resampler.query"], W["resampler.ln_q.weight"], W["resampler.ln_q.bias"], eps)
    Win, bin_ = W["resampler.attn.in_proj_weight"], W["resampler.attn.in_proj_bias"]
    qp = F.linear(q + pos_q, Win[:E], bin_[:E])                              # [Q,E]
    kp = F.linear(x + pos_kv, Win[E:2 * E], bin_[E:2 * E])                   # [B,N,E]
    vp = F.linear(x, Win[2 * E:], bin_[2 * E:])
    qh = qp.reshape(1, -1, nh, hd).permute(0, 2, 1, 3).expand(B, -1, -1, -1)
    kh = kp.reshape(B, N, nh, hd).permute(0, 2, 1, 3)
    vh = vp.reshape(B, N, nh, hd).permute(0, 2, 1, 3)
    a = sdpa(qh, kh, vh).permute(0, 2, 1, 3).reshape(B, -1, E)
    a = F.linear(a, W["resampler.attn.out_proj.weight"], W["resampler.attn.out_proj.bias"])
    a = layer_norm(a, W["resampler.ln_post.weight"], W["resampler.ln_post.bias"], eps)
    out = a @ W["resampler.proj"]
    if taps is not None:
        taps["resampler_out"] = out.clone()
    return out


# --------------------------------------------------------------------------- decoder ---
def rms_norm(x, w, eps):
    # modeling_minicpm.py:119-123 (fp32: the dtype round trip is the identity)
    var = x.pow(2).mean(-1, keepdim=True)
    return x * torch.rsqrt(var + eps) * w


def rope_tables(head_dim: int, L: int, theta: float):
    # modeling_minicpm.py:142-172
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2).float() / head_dim))
    fr = torch.outer(torch.arange(L).float(), inv)
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos(), emb.sin()


def apply_rope(q, k, cos, sin):
    # modeling_minicpm.py:252-290 (rotate_half; position_ids = arange(L))
    def rot(x):
        h = x.shape[-1] // 2
        return torch.cat((-x[..., h:], x[..., :h]), dim=-1)
    return q * cos + rot(q) * sin, k * cos + rot(k) * sin


def decoder_forward(W, cfg, h: torch.Tensor, attn_mask_2d: torch.Tensor, taps=None) -> torch.Tensor:
    """MiniCPMModel.forward with inputs_embeds: modeling_minicpm.py:1147-1304;
    MiniCPMDecoderLayer :939-1004; MiniCPMSdpaAttention :832-910; MiniCPMMLP :293-335.
    Always causal (is_causal flag unused, :374).  h [B,L,E]; attn_mask_2d [B,L] (1 = token)."""
    B, L, E = h.shape
    nh, hd = cfg.num_heads, cfg.head_dim
    cos, sin = rope_tables(hd, L, cfg.rope_theta)
    neg = torch.finfo(torch.float32).min
    causal = torch.full((L, L), neg).triu(1)
    pad = (1.0 - attn_mask_2d.float())[:, None, None, :] * neg               # key padding
    mask = (causal[None, None] + pad).clamp(min=neg)
    rs = cfg.residual_scale
    for n in range(cfg.num_layers):
        p = f"llm.model.layers.{n}."
        x = rms_norm(h, W[p + "input_layernorm.weight"], cfg.rms_norm_eps)
        q = F.linear(x, W[p + "self_attn.q_proj.weight"]).view(B, L, nh, hd).transpose(1, 2)
        k = F.linear(x, W[p + "self_attn.k_proj.weight"]).view(B, L, nh, hd).transpose(1, 2)
        v = F.linear(x, W[p + "self_attn.v_proj.weight"]).view(B, L, nh, hd).transpose(1, 2)
        q, k = apply_rope(q, k, cos, sin)
        a = sdpa(q, k, v, mask=mask).transpose(1, 2).reshape(B, L, E)
        h = h + F.linear(a, W[p + "self_attn.o_proj.weight"]) * rs
        x = rms_norm(h, W[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        m = F.silu(F.linear(x, W[p + "mlp.gate_proj.weight"])) * F.linear(x, W[p + "mlp.up_proj.weight"])
        h = h + F.linear(m, W[p + "mlp.down_proj.weight"]) * rs
        if taps is not None and n == 0:
            taps["dec_layer0"] = h.clone()
    h = rms_norm(h, W["llm.model.norm.weight"], cfg.rms_norm_eps)
    return h


def wmean_pool_normalize(hidden: torch.Tensor, attn_mask_2d: torch.Tensor) -> torch.Tensor:
    """dense_retrieval_model.py:180-184 (wmean) + :222-223 (F.normalize, eps 1e-12)."""
    m = attn_mask_2d.to(torch.int64)
    w = (m * m.cumsum(dim=1)).float()
    s = torch.sum(hidden * w.unsqueeze(-1), dim=1)
    d = w.sum(dim=1, keepdim=True)
    reps = s / d
    return reps / reps.norm(dim=1, keepdim=True).clamp_min(1e-12)


def pool_normalize(hidden: torch.Tensor, attn_mask_2d: torch.Tensor, pooling: str = "wmean") -> torch.Tensor:
    """The deterministic poolings of DRModel.encode (dense_retrieval_model.py:172-220) + F.normalize (:222-223):
    wmean (:180-184), mean (:204-207), lasttoken = last_token_pool with right padding (:26-34,172-177),
    cls = hidden[:, 0] (:217-218).  (drop_wmean / drop_mean / lasttoken_simcse apply dropout in training mode
    even at inference — stochastic in the reference itself — and are not restated.)"""
    m = attn_mask_2d.to(torch.int64)
    if pooling == "wmean":
        return wmean_pool_normalize(hidden, attn_mask_2d)
    if pooling == "mean":
        reps = torch.sum(hidden * m.unsqueeze(-1).float(), dim=1) / m.sum(dim=1, keepdim=True).float()
    elif pooling == "lasttoken":
        reps = hidden[torch.arange(hidden.shape[0]), m.sum(dim=1) - 1]
    elif pooling == "cls":
        reps = hidden[:, 0, :]
    else:
        raise ValueError("Unknown pooling type: {}".format(pooling))
    return reps / reps.norm(dim=1, keepdim=True).clamp_min(1e-12)


# ----------------------------------------------------------------------- whole model ---
def encode(W, cfg, input_ids: Sequence[Sequence[int]], image_bounds: Sequence[Sequence[Tuple[int, int]]],
           pixel_values: Sequence[Sequence[np.ndarray]], taps: Optional[dict] = None, pooling: str = "wmean") -> torch.Tensor:
    """VisRAG_Ret.forward + pooling: modeling_visrag_ret.py:86-126,
    get_vllm_embedding modeling_minicpmv.py:124-171 (embedding * scale_emb, then scatter of
    the un-scaled vision rows into [start, end) of each image bound),
    get_vision_embedding :95-122 (first slice alone, remaining slices as one batch),
    right padding `pad` :440-479, then dense_retrieval_model.py:180-184,222-223.
    input_ids: per item token ids (already truncated); image_bounds: per item [(start,end)];
    pixel_values: per item list of u8 HWC arrays (slices; [] for text).  -> [B,E] f32."""
    with torch.no_grad():
        B = len(input_ids)
        L = max(len(x) for x in input_ids)
        ids = torch.zeros((B, L), dtype=torch.long)
        mask = torch.zeros((B, L), dtype=torch.int64)
        for i, x in enumerate(input_ids):
            ids[i, :len(x)] = torch.tensor(list(x), dtype=torch.long)
            mask[i, :len(x)] = 1
        emb = F.embedding(ids, W["llm.model.embed_tokens.weight"]) * cfg.scale_emb
        P = cfg.patch_size
        for i in range(B):
            pv = pixel_values[i]
            if len(pv) == 0:
                continue
            groups = [[pv[0]]] + ([list(pv[1:])] if len(pv) > 1 else [])
            outs = []
            for grp in groups:
                px = torch.stack([to_pixel_tensor(a) for a in grp])
                tgt = (math.ceil(px.shape[-2] / P), math.ceil(px.shape[-1] / P))
                feats = vit_forward(W, cfg, px, taps if (i == 0 and grp is groups[0]) else None)
                outs.append(resampler_forward(W, cfg, feats, tgt,
                                              taps if (i == 0 and grp is groups[0]) else None))
            vis = torch.cat(outs, dim=0).reshape(-1, cfg.hidden_size)
            rows = torch.cat([torch.arange(s, e) for s, e in image_bounds[i]]) if len(image_bounds[i]) else None
            if rows is not None:
                emb[i, rows] = vis[: len(rows)]
        if taps is not None:
            taps["inputs_embeds"] = emb.clone()
        hidden = decoder_forward(W, cfg, emb, mask, taps)
        if taps is not None:
            taps["last_hidden"] = hidden.clone()
        return pool_normalize(hidden, mask, pooling)


# ------------------------------------------------------------------------- retrieval ---
def search_topk(Q: np.ndarray, C: np.ndarray, k: int):
    """_retrieve_one_shard: dense_retriever.py:13-34 — scores = Q @ C^T (fp32), top-k per
    query.  torch.topk's tie order is unspecified; the oracle fixes it: higher score first,
    lower corpus index first among equal scores.  -> (scores [Nq,k] f32, idx [Nq,k] i64)."""
    S = torch.from_numpy(np.ascontiguousarray(Q, dtype=np.float32)) @ \
        torch.from_numpy(np.ascontiguousarray(C, dtype=np.float32)).T
    S = S.numpy()
    k = min(k, S.shape[1])
    order = np.lexsort((np.broadcast_to(np.arange(S.shape[1]), S.shape), -S), axis=1)[:, :k]
    return np.take_along_axis(S, order, axis=1), order.astype(np.int64)


def retrieve(Q: np.ndarray, qids: List[str], shards: List[Tuple[np.ndarray, List[str]]], k: int
             ) -> Dict[str, Dict[str, float]]:
    """distributed_parallel_retrieve: dense_retriever.py:37-97 — per corpus shard top-k,
    union into {qid: {docid: score}} (up to k * n_shards entries per query)."""
    res: Dict[str, Dict[str, float]] = {q: {} for q in qids}
    for C, ids in shards:
        sc, ix = search_topk(Q, C, k)
        for qi, q in enumerate(qids):
            for s, j in zip(sc[qi], ix[qi]):
                res[q][ids[int(j)]] = float(s)
    return res
