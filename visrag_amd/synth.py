"""Deterministic synthetic VisRAG-Ret weights (there is no checkpoint on the box).

Every tensor is produced by a counter-based integer hash (splitmix64 finaliser) of the
element index and a per-tensor stream id, mapped to a uniform value and rounded to a
bf16-representable number.  Integer arithmetic makes the result bit-identical on any
host CPU and on the GPU (torch int64 ops wrap on both), independent of the torch RNG
implementation, so the CPU oracle, the committed golden fixtures and the HIP path on a
fresh GPU box all see exactly the same "checkpoint".

Key names and shapes follow the HF state dict of `VisRAG_Ret` (SURVEY.md appendix A;
built by MiniCPMV.__init__, modeling_minicpmv.py:32-45; resampler.py:105-131;
timm vision_transformer.py:59-107,125-168; modeling_minicpm.py:293-335,352-409).
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Iterator, Tuple

import torch

from .config import VisRAGRetConfig

_C1 = -0x40A7B892E31B1A47  # 0xBF58476D1CE4E5B9 as signed int64
_C2 = -0x6B2FB644ECCEEE15  # 0x94D049BB133111EB as signed int64
_GOLD = -0x61C8864680B583EB  # 0x9E3779B97F4A7C15 as signed int64
_CHUNK = 1 << 25


def _wrap64(v: int) -> int:
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def _mix64(x: torch.Tensor) -> torch.Tensor:
    x = (x ^ ((x >> 30) & ((1 << 34) - 1))) * _C1
    x = (x ^ ((x >> 27) & ((1 << 37) - 1))) * _C2
    x = x ^ ((x >> 31) & ((1 << 33) - 1))
    return x


def stream_id(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) * 1000003 + seed * 7919 + 12345) & 0x7FFFFFFF


def uniform_pm1(numel: int, stream: int, device="cpu") -> torch.Tensor:
    """numel fp32 values in [-1, 1), 24-bit resolution, from hash(stream, index)."""
    out = torch.empty(numel, dtype=torch.float32, device=device)
    base = _wrap64(stream * _GOLD)
    for lo in range(0, numel, _CHUNK):
        hi = min(numel, lo + _CHUNK)
        idx = torch.arange(lo, hi, dtype=torch.int64, device=device)
        h = _mix64(idx * _GOLD + base)
        u = ((h >> 40) & 0xFFFFFF).to(torch.float32)  # exact: < 2**24
        out[lo:hi] = u * (2.0 / 16777216.0) - 1.0
    return out


def _bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def synth_tensor(name: str, shape, amp: float, seed: int, offset: float = 0.0,
                 device="cpu") -> torch.Tensor:
    """fp32 tensor whose every value is bf16-representable: offset + amp * U[-1,1)."""
    numel = int(math.prod(shape))
    u = uniform_pm1(numel, stream_id(name, seed), device=device)
    w = u * amp
    if offset != 0.0:
        w = w + offset
    return _bf16_round(w).reshape(shape)


def _lin_amp(fan_in: int, gain: float = 1.0) -> float:
    # uniform(-a, a) has std a/sqrt(3); target std = gain/sqrt(fan_in)
    return gain * math.sqrt(3.0 / fan_in)


def weight_specs(cfg: VisRAGRetConfig) -> "OrderedDict[str, Tuple[tuple, float, float]]":
    """name -> (shape, amplitude, offset)."""
    D, F = cfg.vit_dim, cfg.vit_hidden
    E, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    P = cfg.patch_size
    s: "OrderedDict[str, Tuple[tuple, float, float]]" = OrderedDict()
    s["vpm.patch_embed.proj.weight"] = ((D, 3, P, P), _lin_amp(3 * P * P, 2.0), 0.0)
    s["vpm.patch_embed.proj.bias"] = ((D,), 0.1, 0.0)
    s["vpm.pos_embed"] = ((1, cfg.vit_pos_grid ** 2, D), 0.5, 0.0)
    for n in range(cfg.vit_depth):
        p = f"vpm.blocks.{n}."
        s[p + "norm1.weight"] = ((D,), 0.2, 1.0)
        s[p + "norm1.bias"] = ((D,), 0.1, 0.0)
        s[p + "attn.qkv.weight"] = ((3 * D, D), _lin_amp(D, 1.5), 0.0)
        s[p + "attn.qkv.bias"] = ((3 * D,), 0.1, 0.0)
        s[p + "attn.proj.weight"] = ((D, D), _lin_amp(D), 0.0)
        s[p + "attn.proj.bias"] = ((D,), 0.05, 0.0)
        s[p + "norm2.weight"] = ((D,), 0.2, 1.0)
        s[p + "norm2.bias"] = ((D,), 0.1, 0.0)
        s[p + "mlp.fc1.weight"] = ((F, D), _lin_amp(D), 0.0)
        s[p + "mlp.fc1.bias"] = ((F,), 0.1, 0.0)
        s[p + "mlp.fc2.weight"] = ((D, F), _lin_amp(F), 0.0)
        s[p + "mlp.fc2.bias"] = ((D,), 0.05, 0.0)
    s["vpm.norm.weight"] = ((D,), 0.2, 1.0)
    s["vpm.norm.bias"] = ((D,), 0.1, 0.0)
    # resampler (resampler.pos_embed is a fixed 8x8 sincos buffer: computed, not synthesised)
    s["resampler.query"] = ((cfg.query_num, E), 1.0, 0.0)
    s["resampler.kv_proj.weight"] = ((E, D), _lin_amp(D), 0.0)
    s["resampler.attn.in_proj_weight"] = ((3 * E, E), _lin_amp(E, 1.5), 0.0)
    s["resampler.attn.in_proj_bias"] = ((3 * E,), 0.1, 0.0)
    s["resampler.attn.out_proj.weight"] = ((E, E), _lin_amp(E), 0.0)
    s["resampler.attn.out_proj.bias"] = ((E,), 0.05, 0.0)
    for ln in ("ln_q", "ln_kv", "ln_post"):
        s[f"resampler.{ln}.weight"] = ((E,), 0.2, 1.0)
        s[f"resampler.{ln}.bias"] = ((E,), 0.1, 0.0)
    s["resampler.proj"] = ((E, E), _lin_amp(E), 0.0)
    # decoder
    s["llm.model.embed_tokens.weight"] = ((V, E), 0.15, 0.0)
    for n in range(cfg.num_layers):
        p = f"llm.model.layers.{n}."
        s[p + "input_layernorm.weight"] = ((E,), 0.2, 1.0)
        s[p + "self_attn.q_proj.weight"] = ((E, E), _lin_amp(E, 1.5), 0.0)
        s[p + "self_attn.k_proj.weight"] = ((E, E), _lin_amp(E, 1.5), 0.0)
        s[p + "self_attn.v_proj.weight"] = ((E, E), _lin_amp(E), 0.0)
        s[p + "self_attn.o_proj.weight"] = ((E, E), _lin_amp(E, 2.0), 0.0)
        s[p + "post_attention_layernorm.weight"] = ((E,), 0.2, 1.0)
        s[p + "mlp.gate_proj.weight"] = ((I, E), _lin_amp(E, 1.5), 0.0)
        s[p + "mlp.up_proj.weight"] = ((I, E), _lin_amp(E, 1.5), 0.0)
        s[p + "mlp.down_proj.weight"] = ((E, I), _lin_amp(I, 2.0), 0.0)
    s["llm.model.norm.weight"] = ((E,), 0.2, 1.0)
    return s


def iter_synth_weights(cfg: VisRAGRetConfig, seed: int = 0, device="cpu"
                       ) -> Iterator[Tuple[str, torch.Tensor]]:
    for name, (shape, amp, off) in weight_specs(cfg).items():
        yield name, synth_tensor(name, shape, amp, seed, off, device=device)


def synth_state_dict(cfg: VisRAGRetConfig, seed: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    return OrderedDict(iter_synth_weights(cfg, seed, device))


def param_count(cfg: VisRAGRetConfig) -> int:
    return sum(int(math.prod(sh)) for sh, _, _ in weight_specs(cfg).values())


# ---------------------------------------------------------------------------------------
# synthetic inputs (BASELINE.json configs: 448x448 page images + text queries)
# ---------------------------------------------------------------------------------------
def _np_mix64(x):
    import numpy as np
    x = x.astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def synth_pages(n: int, size: int = 448, seed: int = 0, first: int = 0):
    """n structurally diverse uint8 HWC page images (numpy [n,size,size,3]), pages
    `first .. first+n-1` of an endless deterministic corpus.  Integer-hash arithmetic only
    (bit-identical on every host): a coarse random colour mosaic (low-frequency structure),
    dark text-like bars whose layout depends on the page id, and +-7 pixel noise, so that
    pages are separable in embedding space (hard parts)."""
    import numpy as np
    pages = np.empty((n, size, size, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:size, 0:size]
    with np.errstate(over="ignore"):
        for k in range(n):
            pid = np.uint64((first + k) * 1000003 + seed * 7919 + 17)
            def h(i, salt):
                return _np_mix64(np.asarray(i, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
                                 + pid * np.uint64(0xD1B54A32D192ED03) + np.uint64(salt))
            cell = 16 + int(h(0, 1) % np.uint64(5)) * 16          # 16..80 px mosaic cells
            cid = (yy // cell) * 64 + (xx // cell)
            img = np.empty((size, size, 3), dtype=np.int32)
            for c in range(3):
                img[..., c] = 150 + (h(cid, 10 + c) % np.uint64(100)).astype(np.int32)
            nbars = 5 + int(h(0, 2) % np.uint64(36))
            for b in range(nbars):
                y0 = int(h(b, 3) % np.uint64(size - 8)); hh = 3 + int(h(b, 4) % np.uint64(7))
                x0 = int(h(b, 5) % np.uint64(size - 40))
                ww = 20 + int(h(b, 6) % np.uint64(size - x0 - 20))
                dark = 1 + int(h(b, 7) % np.uint64(7))               # keep dark/16 of the value
                img[y0:y0 + hh, x0:x0 + ww, :] = img[y0:y0 + hh, x0:x0 + ww, :] * dark // 16
            noise = (h(yy * size + xx, 8) % np.uint64(15)).astype(np.int32) - 7
            img += noise[..., None]
            pages[k] = np.clip(img, 0, 255).astype(np.uint8)
    return pages


def synth_deck_pages(n_decks: int, per_deck: int = 10, size: int = 448, seed: int = 0, first_deck: int = 0,
                     slide_bars: int = 3, slide_noise: bool = True):
    """`n_decks * per_deck` pages shaped like a slide deck embedded page after page: the slides of a deck share the
    template (colour mosaic + the template's bars, keyed by the deck id) and differ in `slide_bars` text-like bars of
    their own (small: page-number-sized marks) and, with `slide_noise`, the pixel noise (keyed by (deck, slide); else by the deck).  Page order: deck-major.  Integer-hash arithmetic only, like
    `synth_pages`.  Used by the config1sep parity fixture: a query's scores over a deck are near-ties, the decks are as
    far apart as unrelated pages, so the top-`per_deck` cut of a ranking falls BETWEEN decks."""
    import numpy as np
    pages = np.empty((n_decks * per_deck, size, size, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:size, 0:size]
    with np.errstate(over="ignore"):
        for d in range(n_decks):
            did = np.uint64((first_deck + d) * 1000003 + seed * 7919 + 900000017)

            def h(i, salt, pid=did):
                return _np_mix64(np.asarray(i, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
                                 + pid * np.uint64(0xD1B54A32D192ED03) + np.uint64(salt))
            cell = 16 + int(h(0, 1) % np.uint64(5)) * 16
            cid = (yy // cell) * 64 + (xx // cell)
            base = np.empty((size, size, 3), dtype=np.int32)
            for c in range(3):
                base[..., c] = 150 + (h(cid, 10 + c) % np.uint64(100)).astype(np.int32)

            def bars(img, n, pid, salt0, small=False):
                for b in range(n):
                    y0 = int(h(b, salt0 + 3, pid) % np.uint64(size - 8)); hh = 3 + int(h(b, salt0 + 4, pid) % np.uint64(3 if small else 7))
                    x0 = int(h(b, salt0 + 5, pid) % np.uint64(size - 40))
                    ww = 20 + int(h(b, salt0 + 6, pid) % np.uint64(29 if small else size - x0 - 20))
                    dark = (10 if small else 1) + int(h(b, salt0 + 7, pid) % np.uint64(4 if small else 7))
                    img[y0:y0 + hh, x0:x0 + ww, :] = img[y0:y0 + hh, x0:x0 + ww, :] * dark // 16
            bars(base, 5 + int(h(0, 2) % np.uint64(36)), did, 0)
            for j in range(per_deck):
                sid = np.uint64(int(did) * 64 + j + 1)
                img = base.copy()
                bars(img, slide_bars, sid, 100, small=True)        # a slide's own mark: 20-48 x 3-5 px, 19-37 % darker
                noise = (h(yy * size + xx, 8, sid if slide_noise else did) % np.uint64(15)).astype(np.int32) - 7
                img += noise[..., None]
                pages[d * per_deck + j] = np.clip(img, 0, 255).astype(np.uint8)
    return pages


_WORDS = ("revenue table chart figure growth annual report market share total net income "
          "page section summary results method analysis energy policy climate health data "
          "model system network design process quality budget forecast region quarter").split()


def synth_queries(n: int, seed: int = 0, min_words: int = 4, max_words: int = 12):
    import numpy as np
    rng = np.random.default_rng(seed + 1000)
    out = []
    for _ in range(n):
        k = int(rng.integers(min_words, max_words + 1))
        out.append(" ".join(_WORDS[int(j)] for j in rng.integers(0, len(_WORDS), size=k)))
    return out


def synth_pages_gpu(n: int, size: int = 448, seed: int = 0, first: int = 0, device: int = 0, out=None):
    """The same pages as `synth_pages(n, size, seed, first)`, bit for bit, produced on the GPU (vr_synth_pages):
    uint8 cuda tensor [n, size, size, 3].  100 000 distinct pages for BASELINE config 3 take seconds instead of the
    host generator's half hour."""
    import ctypes as C

    from . import _lib
    lib = _lib.load()
    if out is None:
        out = torch.empty((n, size, size, 3), dtype=torch.uint8, device=f"cuda:{device}")
    assert out.is_cuda and out.dtype == torch.uint8 and out.is_contiguous() and out.numel() >= n * size * size * 3
    _lib.check(lib.vr_synth_pages(int(device), C.c_void_p(out.data_ptr()), int(n), int(size), int(seed), int(first),
                                  C.c_void_p(int(torch.cuda.current_stream(int(device)).cuda_stream))), "vr_synth_pages")
    return out[:n]
