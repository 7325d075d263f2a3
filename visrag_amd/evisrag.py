"""Host side of the EVisRAG generator: the call sites of the reference's
`src/evisrag/predict.py` on top of the C ABI of include/visrag_gen.h.

    llm = LLM(model=cfg_or_dir, dtype="bfloat16", limit_mm_per_prompt={"image": 5})            # predict.py:112-117
    sp = SamplingParams(temperature=0.0, repetition_penalty=1.05, max_tokens=2048)              # predict.py:119-123
    outs = llm.generate([{"prompt_token_ids": ids, "multi_modal_data": {...}}], sampling_params=sp)   # predict.py:147
    outs[0].outputs[0].token_ids / .text

`LLM(model=<checkpoint directory>)` reads a HuggingFace Qwen2.5-VL checkpoint the way vLLM does for predict.py:112: config.json
(language model + vision_config), preprocessor_config.json (the image processor's pixel limits and normalisation),
generation_config.json (stop tokens), the *.safetensors shards, and the directory's tokenizer — after which a prompt may be
the chat-templated STRING of predict.py:134-145 (`{"prompt": prompt, "multi_modal_data": {"image": image_inputs}}`).  The
chat template itself stays with the caller's `AutoProcessor`, as in the reference.  `LLM(model=GenConfig(...), weights=...)`
builds a model without a directory (tests, benchmarks: there is no checkpoint on this box); then a prompt is token ids.
Either way the page images travel as:
  * `{"image": [PIL.Image, ...]}` like predict.py:140-145 — with a vision tower attached (`LLM(..., vision=VisionConfig())`)
    the images go through the Qwen2-VL image processing (`process_images`: smart_resize, bicubic resize, normalise,
    patchify) and the HIP tower; every `image_token_id` in the ids stands for one image (the chat template's single
    <|image_pad|>) and is expanded to the image's token count, or the ids arrive already expanded;
  * or precomputed rows: `{"image_embeds": [f32 [h*w, hidden], ...], "image_grids": [(h, w), ...]}` where h x w is the
    MERGED token grid of the image and the ids hold one placeholder per row.
Positions follow the reference model's `get_rope_index`: text advances all three axes together; an image keeps the
temporal axis fixed and runs height / width over its grid; text after it resumes at the largest position so far + 1.
There is no CPU fallback: without libvisrag_hip.so this module raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib


@dataclass
class GenConfig:
    """Language-model dimensions of Qwen2.5-VL-7B (EVisRAG-7B) by default."""
    hidden_size: int = 3584
    num_hidden_layers: int = 28
    num_attention_heads: int = 28
    num_key_value_heads: int = 4
    intermediate_size: int = 18944
    vocab_size: int = 152064
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    mrope_section: Tuple[int, int, int] = (16, 24, 24)
    image_token_id: int = 151655
    eos_token_ids: Tuple[int, ...] = (151645, 151643)


@dataclass
class VisionConfig:
    """Vision-tower dimensions of Qwen2.5-VL-7B by default (HF Qwen2_5_VLVisionConfig names); min/max_pixels are the
    image processor's area limits (preprocessor_config.json of the checkpoint)."""
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
    min_pixels: int = 56 * 56
    max_pixels: int = 28 * 28 * 1280
    image_mean: Tuple[float, float, float] = (0.48145466, 0.4578275, 0.40821073)
    image_std: Tuple[float, float, float] = (0.26862954, 0.26130258, 0.27577711)

    @property
    def patch_dim(self) -> int:
        return self.in_channels * self.temporal_patch_size * self.patch_size * self.patch_size

    def to_c(self, max_rows: int) -> "_lib.VGVisionConfig":
        full = list(self.fullatt_block_indexes)
        if len(full) > 16:
            raise ValueError("at most 16 full-attention blocks")
        return _lib.VGVisionConfig(self.depth, self.hidden_size, self.num_heads, self.intermediate_size, self.out_hidden_size,
                                   self.in_channels, self.patch_size, self.temporal_patch_size, self.spatial_merge_size,
                                   self.window_size, len(full), (C.c_int32 * 16)(*(full + [0] * (16 - len(full)))), int(max_rows),
                                   self.rms_norm_eps)


def remap_checkpoint_key(k: str) -> str:
    """State-dict key of any Qwen2.5-VL checkpoint layout -> the layout vg_load_weight reads ("model.language_model.*",
    "model.visual.*", "lm_head.weight").  Checkpoints written by transformers < 4.52 (EVisRAG's 4.51.3) use
    "model.layers.*" / "model.embed_tokens.*" / "model.norm.*" and "visual.*"."""
    if k.startswith("visual."):
        return "model." + k
    if k.startswith("model.") and not k.startswith(("model.language_model.", "model.visual.")):
        return "model.language_model." + k[len("model."):]
    return k


def read_checkpoint_configs(path: str):
    """(GenConfig, VisionConfig or None, tie_word_embeddings) from a checkpoint directory: config.json in the flat layout
    of transformers 4.51 or with "text_config" / "rope_parameters" of later versions; preprocessor_config.json and
    generation_config.json when present."""
    import json
    import os
    with open(os.path.join(path, "config.json")) as f:
        j = json.load(f)
    t = dict(j)
    t.update(j.get("text_config") or {})
    rope = t.get("rope_parameters") or t.get("rope_scaling") or {}
    g = GenConfig()
    cfg = GenConfig(
        hidden_size=t.get("hidden_size", g.hidden_size), num_hidden_layers=t.get("num_hidden_layers", g.num_hidden_layers),
        num_attention_heads=t.get("num_attention_heads", g.num_attention_heads),
        num_key_value_heads=t.get("num_key_value_heads", g.num_key_value_heads),
        intermediate_size=t.get("intermediate_size", g.intermediate_size), vocab_size=t.get("vocab_size", g.vocab_size),
        rms_norm_eps=t.get("rms_norm_eps", g.rms_norm_eps), rope_theta=rope.get("rope_theta", t.get("rope_theta", g.rope_theta)),
        mrope_section=tuple(rope.get("mrope_section", g.mrope_section)), image_token_id=j.get("image_token_id", g.image_token_id))
    eos = j.get("eos_token_id", t.get("eos_token_id"))
    gpath = os.path.join(path, "generation_config.json")
    if os.path.exists(gpath):
        with open(gpath) as f:
            eos = json.load(f).get("eos_token_id", eos)
    if eos is not None:
        cfg.eos_token_ids = tuple(eos) if isinstance(eos, (list, tuple)) else (int(eos),)
    vision = None
    v = j.get("vision_config")
    if v:
        d = VisionConfig()
        vision = VisionConfig(
            depth=v.get("depth", d.depth), hidden_size=v.get("hidden_size", d.hidden_size), num_heads=v.get("num_heads", d.num_heads),
            intermediate_size=v.get("intermediate_size", d.intermediate_size),
            out_hidden_size=v.get("out_hidden_size", cfg.hidden_size), in_channels=v.get("in_channels", v.get("in_chans", d.in_channels)),
            patch_size=v.get("patch_size", d.patch_size), temporal_patch_size=v.get("temporal_patch_size", d.temporal_patch_size),
            spatial_merge_size=v.get("spatial_merge_size", d.spatial_merge_size), window_size=v.get("window_size", d.window_size),
            fullatt_block_indexes=tuple(v.get("fullatt_block_indexes", d.fullatt_block_indexes)))
        ppath = os.path.join(path, "preprocessor_config.json")
        if os.path.exists(ppath):
            with open(ppath) as f:
                pj = json.load(f)
            size = pj.get("size") or {}
            vision.min_pixels = int(pj.get("min_pixels", size.get("shortest_edge", vision.min_pixels)))
            vision.max_pixels = int(pj.get("max_pixels", size.get("longest_edge", vision.max_pixels)))
            vision.image_mean = tuple(pj.get("image_mean", vision.image_mean))
            vision.image_std = tuple(pj.get("image_std", vision.image_std))
    return cfg, vision, bool(t.get("tie_word_embeddings", j.get("tie_word_embeddings", False)))


def iter_checkpoint_weights(path: str, tie_word_embeddings: bool = False):
    """(key for vg_load_weight, CPU tensor) over the *.safetensors shards of a checkpoint directory; a tied lm_head is
    the embedding table a second time."""
    import os
    from safetensors import safe_open
    files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    have_head = False
    for fn in files:
        with safe_open(os.path.join(path, fn), framework="pt", device="cpu") as f:
            for k in f.keys():
                key = remap_checkpoint_key(k)
                have_head |= key == "lm_head.weight"
                t = f.get_tensor(k)
                yield key, t
                if tie_word_embeddings and key == "model.language_model.embed_tokens.weight":
                    have_head = True
                    yield "lm_head.weight", t
    if not have_head:
        raise KeyError(f"no lm_head.weight under {path} (and tie_word_embeddings is false)")


def load_tokenizer(path: str):
    """The checkpoint directory's tokenizer (what vLLM tokenizes predict.py's prompt string with); None if the directory
    has none."""
    import os
    if not any(os.path.exists(os.path.join(path, f)) for f in ("tokenizer.json", "vocab.json", "tokenizer.model")):
        return None
    from transformers import AutoTokenizer
    return AutoTokenizer.from_pretrained(path, local_files_only=True)


@dataclass
class SamplingParams:       # vllm.SamplingParams as predict.py:119-123 uses it
    temperature: float = 0.1
    repetition_penalty: float = 1.05
    max_tokens: int = 2048
    seed: int = 0
    stop_token_ids: Optional[Sequence[int]] = None


@dataclass
class CompletionOutput:
    token_ids: List[int]
    text: str = ""


@dataclass
class RequestOutput:        # pred.outputs[0].text (predict.py:154)
    outputs: List[CompletionOutput] = field(default_factory=list)
    prompt_token_ids: List[int] = field(default_factory=list)


def gen_weight_specs(cfg: "GenConfig"):
    """HF state-dict keys of the language model (Qwen2_5_VLForConditionalGeneration) -> (shape, amplitude, offset) of
    the deterministic synthetic weights (visrag_amd.synth's counter hash) the tests and benchmarks load: there is no
    checkpoint on the box."""
    import math
    H, KV, I, E = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size, cfg.hidden_size
    hd = E // H
    lin = lambda fan_in, g=1.0: g * math.sqrt(3.0 / fan_in)
    specs = {"model.language_model.embed_tokens.weight": ((cfg.vocab_size, E), 0.05, 0.0)}
    for l in range(cfg.num_hidden_layers):
        p = f"model.language_model.layers.{l}."
        specs[p + "self_attn.q_proj.weight"] = ((H * hd, E), lin(E), 0.0)
        specs[p + "self_attn.q_proj.bias"] = ((H * hd,), 0.1, 0.0)
        specs[p + "self_attn.k_proj.weight"] = ((KV * hd, E), lin(E), 0.0)
        specs[p + "self_attn.k_proj.bias"] = ((KV * hd,), 0.1, 0.0)
        specs[p + "self_attn.v_proj.weight"] = ((KV * hd, E), lin(E), 0.0)
        specs[p + "self_attn.v_proj.bias"] = ((KV * hd,), 0.1, 0.0)
        specs[p + "self_attn.o_proj.weight"] = ((E, H * hd), lin(H * hd, 0.5), 0.0)
        specs[p + "mlp.gate_proj.weight"] = ((I, E), lin(E), 0.0)
        specs[p + "mlp.up_proj.weight"] = ((I, E), lin(E), 0.0)
        specs[p + "mlp.down_proj.weight"] = ((E, I), lin(I, 0.5), 0.0)
        specs[p + "input_layernorm.weight"] = ((E,), 0.1, 1.0)
        specs[p + "post_attention_layernorm.weight"] = ((E,), 0.1, 1.0)
    specs["model.language_model.norm.weight"] = ((E,), 0.1, 1.0)
    specs["lm_head.weight"] = ((cfg.vocab_size, E), lin(E, 2.0), 0.0)
    return specs


def iter_synth_gen_weights(cfg: "GenConfig", seed: int = 0, device="cpu", bf16: bool = False):
    from .synth import synth_tensor
    import torch
    for k, (shape, amp, off) in gen_weight_specs(cfg).items():
        t = synth_tensor(k, shape, amp, seed, off, device=device)
        yield k, (t.to(torch.bfloat16) if bf16 else t)


def bench_generate(n_images: int = 5, answer_tokens: int = 64, queries: int = 2, device: int = 0, vision: bool = True,
                   chain=None, a4_pages: bool = False, a4_tokens: int = 2048) -> dict:
    """EVisRAG-7B-shaped generation (BASELINE config 5: the top retrieved pages go to the generator, one query at a
    time like src/evisrag/predict.py:128-149): random weights of Qwen2.5-VL-7B (vision tower + language model), pages
    as the image processor's pixel rows (448 x 448: 32 x 32 patches -> 16 x 16 image tokens).  Returns vision / prefill
    / decode timings, the queries/s for this answer length and the decode step's weight-streaming rate against the
    HBM.  vision=False: image tokens as precomputed embedding rows, language model only."""
    import itertools
    import time
    import torch
    cfg = GenConfig()
    vc = VisionConfig() if vision else None
    t0 = time.time()
    a4_pages = bool(a4_pages and vision)
    a4_hw = smart_resize(2339, 1654, vc.patch_size * vc.spatial_merge_size, vc.min_pixels, vc.max_pixels) if a4_pages else (0, 0)
    a4_rows = (a4_hw[0] // vc.patch_size) * (a4_hw[1] // vc.patch_size) if a4_pages else 0      # patch rows of one A4 page
    a4_prompt = 120 + 2 * n_images + n_images * (a4_rows // 4 if a4_pages else 0)
    llm = LLM(cfg, limit_mm_per_prompt={"image": max(5, n_images)}, max_model_len=max(4096, a4_prompt + a4_tokens + 64),
              max_prefill=max(2048, a4_prompt + 64), device=device, vision=vc,
              max_vision_rows=max(n_images * 1024, n_images * a4_rows + 256) if vision else None, max_num_seqs=5)
    w = iter_synth_gen_weights(cfg, 0, device=f"cuda:{device}", bf16=True)
    if vision:
        w = itertools.chain(w, iter_synth_vision_weights(vc, 0, device=f"cuda:{device}", bf16=True))
    llm.load_weights(w)
    torch.cuda.synchronize(device)
    t_load = time.time() - t0
    specs = gen_weight_specs(cfg)
    params = sum(int(np.prod(s)) for s, _, _ in specs.values())
    stream = params - cfg.vocab_size * cfg.hidden_size           # the embedding table is gathered, not streamed
    rng = np.random.default_rng(0)
    grid = (16, 16)
    ids = [int(t) for t in rng.integers(1000, 50000, 60)]
    for _ in range(n_images):
        ids += [cfg.image_token_id] * (grid[0] * grid[1]) + [198]
    ids += [int(t) for t in rng.integers(1000, 50000, 60)]
    embs = [(rng.standard_normal((grid[0] * grid[1], cfg.hidden_size)) * 0.05).astype(np.float32) for _ in range(n_images)]
    sp = SamplingParams(temperature=0.1, repetition_penalty=1.05, max_tokens=answer_tokens, stop_token_ids=())
    mm = {"image_embeds": embs, "image_grids": [grid] * n_images}
    vis_ms, vis_tf = None, None
    if vision:
        px = (rng.standard_normal((n_images * 1024, vc.patch_dim))).astype(np.float32)
        thw = np.asarray([(1, 32, 32)] * n_images, dtype=np.int32)
        mm = {"pixel_values": px, "image_grid_thw": thw}
    llm.generate([{"prompt_token_ids": ids, "multi_modal_data": mm}],
                 SamplingParams(temperature=0.1, repetition_penalty=1.05, max_tokens=4, stop_token_ids=()))      # warm-up
    if vision:
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(device); a = time.perf_counter()
            llm.encode_images(px, thw, fetch=False)
            torch.cuda.synchronize(device); ts.append(time.perf_counter() - a)
        vis_ms = float(np.median(ts)) * 1e3
        vspecs = vision_weight_specs(vc)
        R = px.shape[0]
        blk = sum(int(np.prod(sh)) for k, (sh, _, _) in vspecs.items() if ".blocks." in k and k.endswith("weight") and len(sh) == 2)
        mrg = sum(int(np.prod(sh)) for k, (sh, _, _) in vspecs.items() if ".merger.mlp." in k and k.endswith("weight"))
        hd = vc.hidden_size // vc.num_heads
        att = 0.0                                            # QK^T + PV: 4 x rows x keys x hidden per block
        for l in range(vc.depth):
            keys = 1024 if l in vc.fullatt_block_indexes else 64
            att += 4.0 * R * keys * vc.hidden_size
        flop = 2.0 * R * (blk + vc.hidden_size * vc.patch_dim) + 2.0 * (R / 4) * mrg + att
        vis_tf = flop / (vis_ms * 1e-3) / 1e12
    pre, dec, tot = [], [], []
    for _ in range(queries):
        torch.cuda.synchronize(device); a = time.perf_counter()
        if vision:
            pos3, _ = llm.prefill_images(ids, px, thw)
        else:
            pos3 = llm.prefill(ids, embs, [grid] * n_images)
        tok = llm.sample(sp, 0)
        torch.cuda.synchronize(device); b = time.perf_counter()
        nxt = int(pos3.max()) + 1
        llm._continue(tok, nxt, answer_tokens, sp, (), True)          # captured steps, one enqueued ahead
        torch.cuda.synchronize(device); c = time.perf_counter()
        pre.append(b - a); dec.append((c - b) / max(1, answer_tokens - 1)); tot.append(c - a)
    # the same steps driven from the host (one vg_decode + vg_sample pair per token): what the capture saves
    if vision:
        pos3, _ = llm.prefill_images(ids, px, thw)
    else:
        pos3 = llm.prefill(ids, embs, [grid] * n_images)
    tok = llm.sample(sp, 0)
    torch.cuda.synchronize(device); b = time.perf_counter()
    llm._continue(tok, int(pos3.max()) + 1, answer_tokens, sp, (), False)
    torch.cuda.synchronize(device); host_dec = (time.perf_counter() - b) / max(1, answer_tokens - 1)
    # pages as u8 images, image processing on the GPU (what generate() does with PIL pages): tower + prefill again
    pages_ms = None
    if vision:
        pg = [torch.from_numpy(rng.integers(0, 256, size=(448, 448, 3), dtype=np.uint8)).to(f"cuda:{device}") for _ in range(n_images)]
        short = [int(t) for t in rng.integers(1000, 50000, 60)] + [cfg.image_token_id, 198] * n_images + [int(t) for t in rng.integers(1000, 50000, 60)]
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(device); a = time.perf_counter()
            dev_pages, thw2 = process_pages_gpu(pg, vc, device)
            llm.prefill_images(short, dev_pages, thw2)
            llm.sample(sp, 0)
            torch.cuda.synchronize(device); ts.append(time.perf_counter() - a)
        pages_ms = float(np.median(ts)) * 1e3
    # five queries handed to generate() together (max_num_seqs = 5): prefills one after the other, ONE decode step per
    # token for all five (vg_decode_batch) — the throughput form of the same workload
    bat_s = None
    if queries > 0:
        prs = [{"prompt_token_ids": ids, "multi_modal_data": mm}] * 5
        llm.generate(prs, SamplingParams(temperature=0.1, repetition_penalty=1.05, max_tokens=3, stop_token_ids=()))
        torch.cuda.synchronize(device); a = time.perf_counter()
        outs = llm.generate(prs, sp)
        torch.cuda.synchronize(device); bat_s = time.perf_counter() - a
        assert all(len(o.outputs[0].token_ids) == answer_tokens for o in outs)
    e2e = chain(llm) if chain is not None else None        # bench.py: query encode -> search -> page fetch -> generate
    # ---- the reference's operating point (predict.py:112-123: five full pages, max_tokens=2048): A4 pages rasterised at 200 dpi
    #      (1654 x 2339) through smart_resize at the processor's pixel limit, prefill of the ~6.4k-token prompt, 2048 answer tokens
    a4 = None
    if a4_pages:
        pg = [torch.from_numpy(rng.integers(0, 256, size=(2339, 1654, 3), dtype=np.uint8)).to(f"cuda:{device}") for _ in range(n_images)]
        short = [int(t) for t in rng.integers(1000, 50000, 60)] + [cfg.image_token_id, 198] * n_images + [int(t) for t in rng.integers(1000, 50000, 60)]
        sp4 = SamplingParams(temperature=0.1, repetition_penalty=1.05, max_tokens=a4_tokens, stop_token_ids=())
        ts, ds, T4 = [], [], 0
        for rep in range(2):
            torch.cuda.synchronize(device); a = time.perf_counter()
            dev_pages, thw4 = process_pages_gpu(pg, vc, device)
            pos3, ids4 = llm.prefill_images(short, dev_pages, thw4)
            tok = llm.sample(sp4, 0)
            torch.cuda.synchronize(device); b = time.perf_counter()
            T4 = len(ids4)
            n_dec = a4_tokens if rep == 1 else 8
            llm._continue(tok, int(pos3.max()) + 1, n_dec, sp4, (), True)
            torch.cuda.synchronize(device); c_ = time.perf_counter()
            ts.append(b - a); ds.append((c_ - b) / max(1, n_dec - 1))
        p4, d4 = ts[1], ds[1]
        prs = [{"prompt_token_ids": short, "multi_modal_data": {"image": pg}}] * 5
        torch.cuda.synchronize(device); a = time.perf_counter()
        outs = llm.generate(prs, sp4)
        torch.cuda.synchronize(device); bat4 = time.perf_counter() - a
        assert all(len(o.outputs[0].token_ids) == a4_tokens for o in outs)
        img_tok = a4_rows // (vc.spatial_merge_size ** 2)
        # language-model prefill flops: weights 2 x stream x T + causal attention 2 x 2 x T^2 / 2 x hidden per layer
        lm_flop = 2.0 * stream * T4 + 2.0 * cfg.num_hidden_layers * T4 * T4 * cfg.hidden_size
        a4 = {"page": f"A4 @ 200 dpi, 1654 x 2339 u8 -> smart_resize {a4_hw[1]} x {a4_hw[0]} (max_pixels {vc.max_pixels})",
              "image_tokens_per_page": img_tok, "patch_rows_per_page": a4_rows, "prompt_tokens": T4, "answer_tokens": a4_tokens,
              "tower_plus_prefill_ms": round(p4 * 1e3, 1), "prefill_tokens_per_s": round(T4 / p4),
              "language_model_prefill_tflop": round(lm_flop / 1e12, 1),
              "tower_plus_prefill_tflops_lower_bound": round(lm_flop / p4 / 1e12, 1),
              "decode_ms_per_token": round(d4 * 1e3, 3),
              "queries_per_s": round(1.0 / (p4 + (a4_tokens - 1) * d4), 4),
              "queries_per_s_five_together": round(5.0 / bat4, 4),
              "what": "GPU image processing (Pillow-exact resize, rescale / normalise / patchify) -> tower -> prefill -> "
                      f"{a4_tokens} captured decode + sample steps over a {T4}+ row cache; one query at a time and five together"}
    llm.close()
    T, p_s, d_s = len(ids), float(np.median(pre)), float(np.median(dec))
    return {
        "workload": f"Qwen2.5-VL-7B-shaped model, bf16, random weights; prompt {T} tokens ({n_images} pages x 256 image tokens "
                    + ("from the vision tower on 448 x 448 pages (1024 patch rows each, host pixel rows in)" if vision else "as embedding rows")
                    + f" + text), {answer_tokens} answer tokens, temperature 0.1, repetition_penalty 1.05, one query at a time",
        "params_billion": round(params / 1e9, 3), "load_s": round(t_load, 1),
        "vision_ms": round(vis_ms, 2) if vision else None, "vision_tflops": round(vis_tf, 1) if vision else None,
        "prefill_ms": round(p_s * 1e3, 2), "prefill_includes_vision": bool(vision),
        "prefill_ms_from_u8_pages": round(pages_ms, 2) if pages_ms is not None else None, "end_to_end": e2e,
        "prefill_tokens_per_s": round(T / p_s), "prefill_tflops": round(2.0 * stream * T / (p_s - (vis_ms or 0.0) * 1e-3) / 1e12, 1),
        "decode_ms_per_token": round(d_s * 1e3, 3), "decode_tokens_per_s": round(1.0 / d_s, 1),
        "decode_ms_per_token_host_driven": round(host_dec * 1e3, 3),
        "queries_per_s": round(1.0 / float(np.median(tot)), 3),
        "queries_per_s_five_together": round(5.0 / bat_s, 3) if bat_s else None,
        "decode_ms_per_step_five_together": round((bat_s - 5 * p_s) / max(1, answer_tokens - 1) * 1e3, 3) if bat_s else None,
        "queries_per_s_at_2048_tokens": round(1.0 / (p_s + 2047 * d_s), 4),
        "a4_pages": a4,
        "roofline": {"bound": "hbm", "kernel": "decode step (one captured hipGraph): vr::gemm_skinny_kernel (M = 1 weight streaming) + attention + norms + sampling",
                     "achieved": round(stream * 2 / d_s / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(stream * 2 / d_s / 8e12, 4),
                     "bytes_per_token": stream * 2}}


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56, max_pixels: int = 28 * 28 * 1280) -> Tuple[int, int]:
    """The size the Qwen2-VL image processor resizes a page to: both sides multiples of `factor` (patch x merge), the area
    within [min_pixels, max_pixels], the aspect ratio kept as closely as the grid allows."""
    import math
    if max(height, width) / min(height, width) > 200:
        raise ValueError(f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}")
    h = round(height / factor) * factor
    w = round(width / factor) * factor
    if h * w > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h = max(factor, math.floor(height / beta / factor) * factor)
        w = max(factor, math.floor(width / beta / factor) * factor)
    elif h * w < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h = math.ceil(height * beta / factor) * factor
        w = math.ceil(width * beta / factor) * factor
    return h, w


def process_images(images, vc: "VisionConfig") -> Tuple[np.ndarray, np.ndarray]:
    """PIL pages -> (pixel_values f32 [rows][patch_dim], image_grid_thw int32 [n][3]) like the reference's image processor
    (what vLLM runs on predict.py:140's `image_inputs`): RGB, bicubic resize to smart_resize's size, /255, normalise, cut
    into patch rows in merge-block-major order, the still image repeated over the temporal patch."""
    from PIL import Image
    p, m, tp = vc.patch_size, vc.spatial_merge_size, vc.temporal_patch_size
    mean = np.asarray(vc.image_mean, dtype=np.float32)
    std = np.asarray(vc.image_std, dtype=np.float32)
    rows, grids = [], []
    for im in images:
        im = im.convert("RGB")
        H, W = smart_resize(im.height, im.width, p * m, vc.min_pixels, vc.max_pixels)
        x = np.asarray(im.resize((W, H), Image.BICUBIC), dtype=np.float32)
        x = ((x * np.float32(1.0 / 255.0) - mean) / std).transpose(2, 0, 1)                  # [C][H][W]
        gh, gw = H // p, W // p
        x = x.reshape(x.shape[0], gh // m, m, p, gw // m, m, p).transpose(1, 4, 2, 5, 0, 3, 6)   # [gh/m][gw/m][m][m][C][p][p]
        x = np.broadcast_to(x[:, :, :, :, :, None], x.shape[:5] + (tp,) + x.shape[5:])
        rows.append(np.ascontiguousarray(x).reshape(gh * gw, -1))
        grids.append((1, gh, gw))
    return np.ascontiguousarray(np.concatenate(rows), dtype=np.float32), np.asarray(grids, dtype=np.int32)


def process_pages_gpu(images, vc: "VisionConfig", device: int = 0):
    """The device half of process_images: every page (PIL image, u8 HWC array or cuda tensor) -> u8 HWC cuda tensor at
    smart_resize's size (Pillow-exact bicubic on the GPU: vr_resize_bicubic; a page already at that size is only
    uploaded) + image_grid_thw.  Rescale / normalise / patchify happen inside vg_vision_encode_pages."""
    import torch
    from .gpu_resize import resize_bicubic, to_device_u8
    p, m = vc.patch_size, vc.spatial_merge_size
    pages, grids = [], []
    for im in images:
        a = to_device_u8(im, device)                       # pinned staging + asynchronous upload
        h, w = int(a.shape[0]), int(a.shape[1])
        H, W = smart_resize(h, w, p * m, vc.min_pixels, vc.max_pixels)
        t = a if (H, W) == (h, w) else resize_bicubic(a, (W, H), device)
        pages.append(t.contiguous())
        grids.append((1, H // p, W // p))
    return pages, np.asarray(grids, dtype=np.int32)


def vision_weight_specs(vc: "VisionConfig"):
    """HF state-dict keys of the vision tower -> (shape, amplitude, offset) of the synthetic weights (benchmarks)."""
    import math
    H, I, P = vc.hidden_size, vc.intermediate_size, vc.patch_size
    lin = lambda fan_in, g=1.0: g * math.sqrt(3.0 / fan_in)
    pre = "model.visual."
    specs = {pre + "patch_embed.proj.weight": ((H, vc.in_channels, vc.temporal_patch_size, P, P), lin(vc.patch_dim), 0.0)}
    for l in range(vc.depth):
        b = f"{pre}blocks.{l}."
        specs[b + "norm1.weight"] = ((H,), 0.1, 1.0)
        specs[b + "norm2.weight"] = ((H,), 0.1, 1.0)
        specs[b + "attn.qkv.weight"] = ((3 * H, H), lin(H), 0.0)
        specs[b + "attn.qkv.bias"] = ((3 * H,), 0.1, 0.0)
        specs[b + "attn.proj.weight"] = ((H, H), lin(H, 0.5), 0.0)
        specs[b + "attn.proj.bias"] = ((H,), 0.05, 0.0)
        specs[b + "mlp.gate_proj.weight"] = ((I, H), lin(H), 0.0)
        specs[b + "mlp.gate_proj.bias"] = ((I,), 0.1, 0.0)
        specs[b + "mlp.up_proj.weight"] = ((I, H), lin(H), 0.0)
        specs[b + "mlp.up_proj.bias"] = ((I,), 0.1, 0.0)
        specs[b + "mlp.down_proj.weight"] = ((H, I), lin(I, 0.5), 0.0)
        specs[b + "mlp.down_proj.bias"] = ((H,), 0.05, 0.0)
    M = H * vc.spatial_merge_size ** 2
    specs[pre + "merger.ln_q.weight"] = ((H,), 0.1, 1.0)
    specs[pre + "merger.mlp.0.weight"] = ((M, M), lin(M), 0.0)
    specs[pre + "merger.mlp.0.bias"] = ((M,), 0.1, 0.0)
    specs[pre + "merger.mlp.2.weight"] = ((vc.out_hidden_size, M), lin(M, 0.25), 0.0)
    specs[pre + "merger.mlp.2.bias"] = ((vc.out_hidden_size,), 0.02, 0.0)
    return specs


def iter_synth_vision_weights(vc: "VisionConfig", seed: int = 0, device="cpu", bf16: bool = False):
    from .synth import synth_tensor
    import torch
    for k, (shape, amp, off) in vision_weight_specs(vc).items():
        t = synth_tensor(k, shape, amp, seed, off, device=device)
        yield k, (t.to(torch.bfloat16) if bf16 else t)


def vision_plan(vc: "VisionConfig", grid_thw) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(order [tokens], window row boundaries, hw [rows][2]) the tower uses for these grids — host-only entry point of the
    library (vg_vision_plan), for tests."""
    g = np.ascontiguousarray(grid_thw, dtype=np.int32).reshape(-1, 3)
    rows = int((g[:, 0] * g[:, 1] * g[:, 2]).sum())
    tokens = rows // vc.spatial_merge_size ** 2
    order = np.zeros(tokens, np.int32)
    bounds = np.zeros(tokens + 1, np.int32)
    hw = np.zeros((rows, 2), np.int32)
    nw = C.c_int32()
    cfg = vc.to_c(max(rows, 1))
    _lib.check(_lib.load().vg_vision_plan(C.byref(cfg), C.c_void_p(g.ctypes.data), len(g), C.c_void_p(order.ctypes.data),
                                          C.c_void_p(bounds.ctypes.data), C.byref(nw), C.c_void_p(hw.ctypes.data)), "vg_vision_plan")
    return order, bounds[:nw.value + 1], hw


def rope_index(ids: Sequence[int], image_token_id: int, grids: Sequence[Tuple[int, int]]) -> np.ndarray:
    """[3][T] temporal / height / width positions of a prompt whose images are runs of `image_token_id`, one run of
    h*w placeholders per (h, w) in `grids`, in order (Qwen2.5-VL get_rope_index for still images)."""
    T = len(ids)
    pos = np.zeros((3, T), dtype=np.int32)
    nxt, i, g = 0, 0, 0
    while i < T:
        if ids[i] == image_token_id:
            if g >= len(grids):
                raise ValueError("more image placeholder runs than image grids")
            h, w = grids[g]
            n = h * w
            if i + n > T or any(t != image_token_id for t in ids[i:i + n]):
                raise ValueError(f"image {g}: expected {n} consecutive placeholders at position {i}")
            pos[0, i:i + n] = nxt
            pos[1, i:i + n] = nxt + np.repeat(np.arange(h), w)
            pos[2, i:i + n] = nxt + np.tile(np.arange(w), h)
            nxt += max(h, w)
            i += n
            g += 1
        else:
            pos[:, i] = nxt
            nxt += 1
            i += 1
    if g != len(grids):
        raise ValueError("fewer image placeholder runs than image grids")
    return pos


class LLM:
    """`vllm.LLM` as the reference constructs it (predict.py:112-117): one model on one GPU, bf16, up to
    `limit_mm_per_prompt["image"]` images per prompt; `generate` takes one prompt at a time like predict.py:128-149."""

    def __init__(self, model, tensor_parallel_size: int = 1, dtype: str = "bfloat16",
                 limit_mm_per_prompt: Optional[Dict[str, int]] = None, max_model_len: Optional[int] = None,
                 max_prefill: Optional[int] = None, max_num_seqs: int = 1,
                 device: int = 0, weights=None, detokenize: Optional[Callable[[List[int]], str]] = None,
                 vision: Optional[VisionConfig] = None, max_vision_rows: Optional[int] = None):
        if tensor_parallel_size != 1:
            raise ValueError("one GPU per model (predict.py:114 uses tensor_parallel_size=1)")
        if dtype not in ("bfloat16", "bf16"):
            raise ValueError("the generator computes in bf16 (predict.py:115)")
        self.tokenizer = None
        # Lengths.  vLLM sizes itself from the checkpoint (128k context); here the KV cache and the prefill workspace are
        # allocated up front: a checkpoint directory gets room for predict.py's workload — five pages at the
        # processor's largest size are 5 x 3 333 image tokens — a bare GenConfig (tests, benchmarks) stays small.
        # A prompt beyond either limit is refused by generate() BEFORE the tower runs.
        from_dir = isinstance(model, str)
        if max_model_len is None:
            max_model_len = 32768 if from_dir else 8192
        if max_prefill is None:
            max_prefill = min(max_model_len, 24576 if from_dir else 4096)
        if isinstance(model, str):                     # a checkpoint directory, like predict.py:112's model_path
            model_dir = model
            model, ck_vision, tied = read_checkpoint_configs(model_dir)
            vision = vision or ck_vision
            if weights is None:
                weights = iter_checkpoint_weights(model_dir, tied)
            self.tokenizer = load_tokenizer(model_dir)
            if detokenize is None and self.tokenizer is not None:
                detokenize = lambda ids, _t=self.tokenizer: _t.decode(ids, skip_special_tokens=True)       # noqa: E731  (vLLM's default)
        if not isinstance(model, GenConfig):
            raise TypeError("model: a checkpoint directory or a GenConfig (+ weights=iterable of (name, tensor))")
        self.cfg: GenConfig = model
        self.max_images = (limit_mm_per_prompt or {"image": 5}).get("image", 5)
        self.gpu_images = True          # generate(): PIL pages are resized / normalised / patchified on the GPU
        self.device = int(device)
        self.detokenize = detokenize
        self.max_model_len, self.max_prefill = int(max_model_len), int(min(max_prefill, max_model_len))
        self._lib = _lib.load()
        c = self.cfg
        # max_num_seqs (vLLM's name): sequences decoded together when generate() is handed several prompts — every weight
        # matrix is streamed once per step for all of them (1 = one at a time, like predict.py's loop; at most 16)
        self.max_num_seqs = max(1, min(16, int(max_num_seqs)))
        vc = _lib.VGConfig(c.hidden_size, c.num_hidden_layers, c.num_attention_heads, c.num_key_value_heads, c.intermediate_size,
                           c.vocab_size, self.max_model_len, self.max_prefill, c.rms_norm_eps, c.rope_theta,
                           (C.c_int32 * 3)(*c.mrope_section), self.max_num_seqs)
        h = C.c_void_p()
        _lib.check(self._lib.vg_create(self.device, C.byref(vc), C.byref(h)), "vg_create")
        self._h = h
        self.vision = vision
        if vision is not None:
            if vision.out_hidden_size != c.hidden_size:
                raise ValueError("vision.out_hidden_size must equal the language model's hidden_size")
            m2 = vision.spatial_merge_size ** 2
            # default: every image at the processor's largest size, capped by what a prefill can take
            rows = max_vision_rows or min(self.max_images * (vision.max_pixels // vision.patch_size ** 2), self.max_prefill * m2)
            self.max_vision_rows = rows // m2 * m2
            vcfg = vision.to_c(self.max_vision_rows)
            _lib.check(self._lib.vg_vision_create(self._h, C.byref(vcfg)), "vg_vision_create")
        if weights is not None:
            self.load_weights(weights)

    def load_weights(self, weights) -> None:
        """weights: iterable of (HF state-dict key, torch tensor f32/bf16 on the CPU or on this GPU)."""
        import torch
        for name, t in (weights.items() if hasattr(weights, "items") else weights):
            name = remap_checkpoint_key(name)
            if t.dtype not in (torch.float32, torch.bfloat16):
                t = t.float()
            t = t.contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            _lib.check(self._lib.vg_load_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), shape, t.dim(),
                                                _lib.VR_DTYPE_BF16 if t.dtype == torch.bfloat16 else _lib.VR_DTYPE_F32,
                                                1 if t.is_cuda else 0), f"vg_load_weight({name})")
        _lib.check(self._lib.vg_finalize(self._h), "vg_finalize")

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.vg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- images ------------------------------------------------------------------------------------------
    def encode_images(self, pixel_values: np.ndarray, image_grid_thw: np.ndarray, fetch: bool = True) -> Optional[np.ndarray]:
        """The vision tower on the processor's output: [tokens][hidden] embedding rows, image by image (also kept on the
        device for the next prefill)."""
        if self.vision is None:
            raise RuntimeError("this LLM has no vision tower (LLM(..., vision=VisionConfig()))")
        px = np.ascontiguousarray(pixel_values, dtype=np.float32)
        g = np.ascontiguousarray(image_grid_thw, dtype=np.int32).reshape(-1, 3)
        rows = int((g[:, 0] * g[:, 1] * g[:, 2]).sum())
        if px.shape != (rows, self.vision.patch_dim):
            raise ValueError(f"pixel_values {px.shape} for grids that hold {rows} x {self.vision.patch_dim}")
        out = np.empty((rows // self.vision.spatial_merge_size ** 2, self.cfg.hidden_size), dtype=np.float32) if fetch else None
        _lib.check(self._lib.vg_vision_encode(self._h, C.c_void_p(px.ctypes.data), C.c_void_p(g.ctypes.data), len(g),
                                              C.c_void_p(out.ctypes.data) if fetch else None, None), "vg_vision_encode")
        return out

    def encode_pages(self, pages, image_grid_thw: np.ndarray, fetch: bool = True) -> Optional[np.ndarray]:
        """The tower on u8 pages resident on the device (process_pages_gpu): the processor's rescale / normalise / patchify
        run inside the library (vg_vision_encode_pages)."""
        if self.vision is None:
            raise RuntimeError("this LLM has no vision tower (LLM(..., vision=VisionConfig()))")
        g = np.ascontiguousarray(image_grid_thw, dtype=np.int32).reshape(-1, 3)
        p = self.vision.patch_size
        for t, (_, gh, gw) in zip(pages, g):
            if tuple(t.shape) != (gh * p, gw * p, 3) or not t.is_cuda or t.dtype != __import__("torch").uint8:
                raise ValueError(f"page {tuple(t.shape)} does not match its grid {gh} x {gw} (u8 HWC cuda tensors)")
        rows = int((g[:, 0] * g[:, 1] * g[:, 2]).sum())
        ptrs = (C.c_void_p * len(pages))(*[t.data_ptr() for t in pages])
        mean = (C.c_float * 3)(*self.vision.image_mean)
        std = (C.c_float * 3)(*self.vision.image_std)
        out = np.empty((rows // self.vision.spatial_merge_size ** 2, self.cfg.hidden_size), dtype=np.float32) if fetch else None
        import torch
        stream = C.c_void_p(int(torch.cuda.current_stream(self.device).cuda_stream))
        _lib.check(self._lib.vg_vision_encode_pages(self._h, ptrs, 1, mean, std, C.c_void_p(g.ctypes.data), len(g),
                                                    C.c_void_p(out.ctypes.data) if fetch else None, stream), "vg_vision_encode_pages")
        return out

    def expand_image_tokens(self, ids: Sequence[int], token_counts: Sequence[int]) -> List[int]:
        """One placeholder per image (the chat template's <|image_pad|>) -> one per image token, like the processor does;
        ids that already hold sum(token_counts) placeholders pass through."""
        tid = self.cfg.image_token_id
        n = sum(1 for t in ids if t == tid)
        if n == sum(token_counts):
            return list(ids)
        if n != len(token_counts):
            raise ValueError(f"{n} image placeholders for {len(token_counts)} images ({sum(token_counts)} image tokens)")
        out, k = [], 0
        for t in ids:
            if t == tid:
                out += [tid] * token_counts[k]
                k += 1
            else:
                out.append(t)
        return out

    def prefill_images(self, ids: Sequence[int], pixel_values, image_grid_thw: np.ndarray) -> Tuple[np.ndarray, List[int]]:
        """Tower + prefill with the embedding rows staying on the device.  Returns (positions, expanded ids).
        pixel_values: the processor's f32 rows, or a list of u8 cuda pages (process_pages_gpu)."""
        c, m = self.cfg, self.vision.spatial_merge_size
        g = np.ascontiguousarray(image_grid_thw, dtype=np.int32).reshape(-1, 3)
        if len(g) > self.max_images:
            raise ValueError(f"{len(g)} images (limit {self.max_images})")
        grids = [(int(h) // m, int(w) // m) for _, h, w in g]
        if any(int(t) != 1 for t, _, _ in g):
            raise ValueError("still images only (predict.py:116 sets the video limit to 0)")
        ids = self.expand_image_tokens(ids, [h * w for h, w in grids])
        if isinstance(pixel_values, (list, tuple)):
            self.encode_pages(pixel_values, g, fetch=False)
        else:
            self.encode_images(pixel_values, g, fetch=False)
        ids_a = np.ascontiguousarray(ids, dtype=np.int32)
        pos3 = np.ascontiguousarray(rope_index(list(ids_a), c.image_token_id, grids), dtype=np.int32)
        rows = np.nonzero(ids_a == c.image_token_id)[0].astype(np.int32)
        _lib.check(self._lib.vg_prefill(self._h, C.c_void_p(ids_a.ctypes.data), len(ids_a), C.c_void_p(rows.ctypes.data), None,
                                        len(rows), C.c_void_p(pos3.ctypes.data), None), "vg_prefill")
        return pos3, ids

    # ---- one sequence ------------------------------------------------------------------------------------
    def prefill(self, ids: Sequence[int], image_embeds: Sequence[np.ndarray] = (), image_grids: Sequence[Tuple[int, int]] = (),
                pos3: Optional[np.ndarray] = None) -> np.ndarray:
        c = self.cfg
        if len(image_embeds) != len(image_grids) or len(image_embeds) > self.max_images:
            raise ValueError(f"{len(image_embeds)} images / {len(image_grids)} grids (limit {self.max_images})")
        ids_a = np.ascontiguousarray(ids, dtype=np.int32)
        if pos3 is None:
            pos3 = rope_index(list(ids_a), c.image_token_id, image_grids)
        pos3 = np.ascontiguousarray(pos3, dtype=np.int32)
        rows = np.nonzero(ids_a == c.image_token_id)[0].astype(np.int32) if len(image_embeds) else np.zeros(0, np.int32)
        emb = (np.ascontiguousarray(np.concatenate([np.asarray(e, dtype=np.float32).reshape(-1, c.hidden_size) for e in image_embeds]))
               if len(image_embeds) else np.zeros((0, c.hidden_size), np.float32))
        if emb.shape[0] != rows.shape[0]:
            raise ValueError(f"{emb.shape[0]} image embedding rows for {rows.shape[0]} placeholders")
        _lib.check(self._lib.vg_prefill(self._h, C.c_void_p(ids_a.ctypes.data), len(ids_a),
                                        C.c_void_p(rows.ctypes.data) if len(rows) else None,
                                        C.c_void_p(emb.ctypes.data) if len(rows) else None, len(rows),
                                        C.c_void_p(pos3.ctypes.data), None), "vg_prefill")
        return pos3

    def logits(self) -> np.ndarray:
        out = np.empty(self.cfg.vocab_size, dtype=np.float32)
        _lib.check(self._lib.vg_logits(self._h, C.c_void_p(out.ctypes.data), None), "vg_logits")
        return out

    def sample(self, sp: SamplingParams, step: int) -> int:
        tok = C.c_int32()
        _lib.check(self._lib.vg_sample(self._h, float(sp.temperature), float(sp.repetition_penalty), int(sp.seed), int(step),
                                       C.byref(tok), None), "vg_sample")
        return int(tok.value)

    # ---- several sequences (vg_select / vg_decode_batch / vg_sample_batch) ---------------------------------------------
    def select(self, slot: int) -> None:
        _lib.check(self._lib.vg_select(self._h, int(slot)), "vg_select")

    def decode_batch(self, slots: Sequence[int], tokens: Sequence[int], positions: Sequence[int]) -> None:
        n = len(slots)
        sl = (C.c_int32 * n)(*[int(x) for x in slots])
        tk = (C.c_int32 * n)(*[int(x) for x in tokens])
        ps = (C.c_int32 * (3 * n))(*[int(p) for p in positions for _ in range(3)])
        _lib.check(self._lib.vg_decode_batch(self._h, n, sl, tk, ps, None), "vg_decode_batch")

    def sample_batch(self, slots: Sequence[int], sp: SamplingParams, step: int) -> List[int]:
        n = len(slots)
        sl = (C.c_int32 * n)(*[int(x) for x in slots])
        out = (C.c_int32 * n)()
        _lib.check(self._lib.vg_sample_batch(self._h, n, sl, float(sp.temperature), float(sp.repetition_penalty), int(sp.seed),
                                             int(step), out, None), "vg_sample_batch")
        return [int(t) for t in out]

    def decode(self, token: int, position: int) -> None:
        p = (C.c_int32 * 3)(position, position, position)
        _lib.check(self._lib.vg_decode(self._h, int(token), p, None), "vg_decode")

    # ---- free-running steps (vg_run_*): no host round trip per token -------------------------------------------
    def run_begin(self, position: int, sp: SamplingParams, first_step: int = 1) -> None:
        _lib.check(self._lib.vg_run_begin(self._h, int(position), float(sp.temperature), float(sp.repetition_penalty), int(sp.seed),
                                          int(first_step), None), "vg_run_begin")

    def run_step(self) -> None:
        _lib.check(self._lib.vg_run_step(self._h), "vg_run_step")

    def run_token(self, index: int) -> int:
        tok = C.c_int32()
        _lib.check(self._lib.vg_run_token(self._h, int(index), C.byref(tok)), "vg_run_token")
        return int(tok.value)

    def run_end(self) -> None:
        _lib.check(self._lib.vg_run_end(self._h), "vg_run_end")

    def _continue(self, first: int, nxt: int, limit: int, sp: SamplingParams, stops, pipelined: bool) -> List[int]:
        """The tokens after the first one.  pipelined: captured decode + sample steps, one step enqueued ahead of the
        token being inspected (a stop token costs at most one unused step); else one vg_decode / vg_sample pair per token."""
        toks = [first]
        if first in stops or limit <= 1:
            return toks
        if not pipelined:
            for step in range(1, limit):
                self.decode(toks[-1], nxt)
                nxt += 1
                toks.append(self.sample(sp, step))
                if toks[-1] in stops:
                    break
            return toks
        self.run_begin(nxt, sp, 1)
        issued = got = 0
        self.run_step(); issued += 1
        while got < issued:
            if issued < limit - 1:
                self.run_step(); issued += 1
            toks.append(self.run_token(got)); got += 1
            if toks[-1] in stops:
                break
        self.run_end()
        return toks

    def _check_lengths(self, n_tokens: int, n_vision_rows: int) -> None:
        """Refuse a prompt the allocations cannot hold, up front and in the caller's terms (the C ABI would answer
        VR_ERR_CAPACITY after the tower has already run)."""
        if n_tokens > self.max_prefill or n_tokens >= self.max_model_len:
            raise ValueError(f"prompt of {n_tokens} tokens ({n_vision_rows // 4 if n_vision_rows else 0} of them image tokens) exceeds "
                             f"max_prefill={self.max_prefill} / max_model_len={self.max_model_len}: construct "
                             "LLM(..., max_model_len=..., max_prefill=...) for it")
        if n_vision_rows and n_vision_rows > getattr(self, "max_vision_rows", 0):
            raise ValueError(f"{n_vision_rows} patch rows exceed the tower's workspace (max_vision_rows={self.max_vision_rows}): "
                             "construct LLM(..., max_vision_rows=...) or lower the processor's max_pixels")

    # ---- predict.py:147 ------------------------------------------------------------------------------------
    def generate(self, prompts, sampling_params: Optional[SamplingParams] = None, pipelined: bool = True) -> List[RequestOutput]:
        sp = sampling_params or SamplingParams()
        stops = set(sp.stop_token_ids if sp.stop_token_ids is not None else self.cfg.eos_token_ids)
        if self.max_num_seqs > 1 and len(prompts) > 1:
            return self._generate_batched(list(prompts), sp, stops)
        outs = []
        for pr in prompts:
            ids, pos3 = self._prefill_prompt(pr)
            nxt = int(pos3.max()) + 1
            limit = min(sp.max_tokens, self.max_model_len - len(ids))
            toks: List[int] = self._continue(self.sample(sp, 0), nxt, limit, sp, stops, pipelined) if limit > 0 else []
            text = self.detokenize(toks) if self.detokenize else ""
            outs.append(RequestOutput([CompletionOutput(toks, text)], ids))
        return outs

    def _generate_batched(self, prompts, sp: SamplingParams, stops) -> List[RequestOutput]:
        """Groups of up to max_num_seqs prompts: every prompt is prefilled into its own slot (tower + prefill one after the
        other, as for a single prompt), then ONE decode step per token serves the whole group (vg_decode_batch: the
        weights are streamed once for all rows); a sequence that has stopped leaves the group.  Rows are independent: a
        prompt's tokens are the ones it gets alone (same seed and step index per sequence)."""
        outs: List[Optional[RequestOutput]] = [None] * len(prompts)
        for g0 in range(0, len(prompts), self.max_num_seqs):
            group = prompts[g0:g0 + self.max_num_seqs]
            ids_l, nxt, toks, limit, live = [], [], [], [], []
            for slot, pr in enumerate(group):
                self.select(slot)
                ids, pos3 = self._prefill_prompt(pr)
                ids_l.append(ids); nxt.append(int(pos3.max()) + 1)
                limit.append(min(sp.max_tokens, self.max_model_len - len(ids)))
                toks.append([self.sample(sp, 0)] if limit[-1] > 0 else [])
                if limit[-1] > 1 and toks[-1][0] not in stops:
                    live.append(slot)
            step = 1
            while live:
                self.decode_batch(live, [toks[sl][-1] for sl in live], [nxt[sl] for sl in live])
                new = self.sample_batch(live, sp, step)
                still = []
                for sl, t in zip(live, new):
                    nxt[sl] += 1
                    toks[sl].append(t)
                    if t not in stops and len(toks[sl]) < limit[sl]:
                        still.append(sl)
                live = still
                step += 1
            self.select(0)
            for slot in range(len(group)):
                text = self.detokenize(toks[slot]) if self.detokenize else ""
                outs[g0 + slot] = RequestOutput([CompletionOutput(toks[slot], text)], ids_l[slot])
        return outs

    def _prefill_prompt(self, pr) -> Tuple[List[int], np.ndarray]:
        """One prompt of generate() into the current slot: tokenise if needed, run the tower on its pages, prefill.
        Returns (prompt ids with the image placeholders expanded, positions)."""
        if pr.get("prompt_token_ids") is not None:
            ids = list(pr["prompt_token_ids"])
        elif self.tokenizer is not None:        # predict.py:143: the chat-templated string
            ids = list(self.tokenizer(pr["prompt"])["input_ids"])
        else:
            raise ValueError("a text prompt needs the checkpoint's tokenizer: LLM(model=<checkpoint directory>), or pass prompt_token_ids")
        mm = pr.get("multi_modal_data") or {}
        images = mm.get("image")
        if images is not None and len(images) == 0:
            images = None                         # a query with no retrieved page: a text-only prompt, like vLLM
        if images is not None or mm.get("pixel_values") is not None:
            if self.vision is None:
                raise RuntimeError("images need a vision tower: LLM(..., vision=VisionConfig())")
            # PIL pages: resize on the GPU (Pillow-exact), rescale / normalise / patchify inside the tower call;
            # processor output handed in by the caller (pixel_values) is used as it is
            px, grid = (mm["pixel_values"], mm["image_grid_thw"]) if mm.get("pixel_values") is not None \
                else (process_pages_gpu(images, self.vision, self.device) if self.gpu_images else process_images(images, self.vision))
            m2 = self.vision.spatial_merge_size ** 2
            g = np.asarray(grid).reshape(-1, 3)
            n_tok = int((g[:, 0] * g[:, 1] * g[:, 2]).sum()) // m2
            n_ph = sum(1 for t in ids if t == self.cfg.image_token_id)
            total = len(ids) - n_ph + n_tok if n_ph == len(g) else len(ids)
            self._check_lengths(total, n_tok * m2)
            pos3, ids = self.prefill_images(ids, px, grid)
        else:
            self._check_lengths(len(ids), 0)
            pos3 = self.prefill(ids, mm.get("image_embeds", ()), mm.get("image_grids", ()), pr.get("positions"))
        return ids, pos3
