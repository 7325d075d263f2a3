"""CPU emulation of the experimental LayerNorm folding (csrc/ln_fold.hip, gemm256w_kernel.h LNF forms) at the ViT's full
width: where do the bf16 roundings fall in the two routes, and what do they cost against the fp32 oracle?

  default:  y = bf16(LayerNorm_fp32(x));              out = bf16(epi(y W^T + bias))          (W bf16, fp32 accumulation)
  folded:   xg = bf16(x o gamma)  (W as it is);       out = bf16(epi(rstd (xg W^T) - rstd mean c1 + c2)),  c1 = W gamma

Six SigLIP blocks (full width 1152 / 4304, 16 heads) on one synthetic 448 x 448 page, both routes carried through the
blocks with their own residual streams, against oracle.vit_forward in fp32 on the same weights.  The folded route must
stay within 1.25x the default route's error — the bar the GPU tests (tests/test_gpu_ln_fold.py) will hold the kernels to
once they have run.  Not a test of the HIP code: a check that the ALGEBRA and its rounding points are sound before GPU
minutes are spent on it."""
import dataclasses

import numpy as np
import torch
import torch.nn.functional as F

from oracle import visrag_ret_oracle as O
from visrag_amd.config import full_config
from visrag_amd.synth import synth_pages, synth_state_dict


def _bf(t):
    return t.to(torch.bfloat16).float()


def _emulate(W, cfg, pixels, folded):
    P, D, H = cfg.patch_size, cfg.vit_dim, cfg.vit_heads
    x = F.conv2d(_bf(pixels), _bf(W["vpm.patch_embed.proj.weight"]), W["vpm.patch_embed.proj.bias"], stride=P)
    gh, gw = x.shape[-2:]
    x = x.permute(0, 2, 3, 1).reshape(1, gh * gw, D) + O.resample_abs_pos_embed(W["vpm.pos_embed"], (gh, gw))
    hd = D // H

    def ln_linear(x, pre, lin, first):
        g, b = W[pre + ".weight"], W[pre + ".bias"]
        w, bias = _bf(W[lin + ".weight"]), W[lin + ".bias"]
        if not folded or first:                                  # (block 0's norm1 stays a LayerNorm launch)
            return _bf(O.layer_norm(x, g, b, cfg.vit_ln_eps)) @ w.T + bias
        mu = x.mean(-1, keepdim=True)
        rstd = 1.0 / torch.sqrt(((x * x).mean(-1, keepdim=True) - mu * mu).clamp_min(0) + cfg.vit_ln_eps)   # (sum, sum of squares: like the kernel)
        c1, c2 = w @ g, bias + w @ b
        return rstd * (_bf(x * g) @ w.T) - rstd * mu * c1 + c2

    for n in range(cfg.vit_depth):
        p = f"vpm.blocks.{n}."
        qkv = _bf(ln_linear(x, p + "norm1", p + "attn.qkv", n == 0)).reshape(1, -1, 3, H, hd).permute(2, 0, 3, 1, 4)
        a = _bf(O.sdpa(qkv[0], qkv[1], qkv[2]).transpose(1, 2).reshape(1, -1, D))
        x = x + a @ _bf(W[p + "attn.proj.weight"]).T + W[p + "attn.proj.bias"]
        y = _bf(F.gelu(ln_linear(x, p + "norm2", p + "mlp.fc1", False)))
        x = x + y @ _bf(W[p + "mlp.fc2.weight"]).T + W[p + "mlp.fc2.bias"]
    return O.layer_norm(x, W["vpm.norm.weight"], W["vpm.norm.bias"], cfg.vit_ln_eps), x


def test_folded_layernorm_rounding_points_cost_no_more_than_the_default_route():
    torch.manual_seed(0)
    cfg = dataclasses.replace(full_config(), vit_depth=6)
    W = {k: v.float() for k, v in synth_state_dict(cfg, 0).items() if k.startswith("vpm.")}
    pixels = O.to_pixel_tensor(synth_pages(1, size=448, seed=0)[0])[None]
    with torch.no_grad():
        ref = O.vit_forward({k: (_bf(v) if v.dim() > 1 and "pos_embed" not in k else v) for k, v in W.items()}, cfg, _bf(pixels))
        (d_out, d_x), (f_out, f_x) = _emulate(W, cfg, pixels, False), _emulate(W, cfg, pixels, True)
    scale = float(ref.abs().max())
    e_def, e_fold = float((d_out - ref).abs().max()) / scale, float((f_out - ref).abs().max()) / scale
    cos = lambda a, b: float(F.cosine_similarity(a[0], b[0], dim=-1).min())
    c_def, c_fold = cos(d_out, ref), cos(f_out, ref)
    # what the residual stream looks like at the point of folding: mean against spread per row
    mu, sd = f_x.mean(-1), f_x.std(-1)
    print(f"default route: max err {e_def:.2e} of scale, min cosine {c_def:.6f};  folded: {e_fold:.2e}, {c_fold:.6f};  "
          f"|row mean| / row std: median {float((mu.abs() / sd).median()):.3f} max {float((mu.abs() / sd).max()):.3f}")
    assert np.isfinite(e_fold)
    assert e_fold <= 1.25 * e_def + 1e-4, (e_fold, e_def)
    assert 1 - c_fold <= 1.5 * (1 - c_def) + 1e-6, (c_fold, c_def)
