"""Host side of the EVisRAG generator: the call sites of the reference's
`src/evisrag/predict.py` on top of the C ABI of include/visrag_gen.h.

    llm = LLM(model=cfg_or_dir, dtype="bfloat16", limit_mm_per_prompt={"image": 5})            # predict.py:112-117
    sp = SamplingParams(temperature=0.0, repetition_penalty=1.05, max_tokens=2048)              # predict.py:119-123
    outs = llm.generate([{"prompt_token_ids": ids, "multi_modal_data": {...}}], sampling_params=sp)   # predict.py:147
    outs[0].outputs[0].token_ids / .text

What this module does NOT contain: the Qwen2.5-VL vision tower and the tokenizer / chat template (the reference gets
them from the checkpoint directory through `AutoProcessor`; there is no checkpoint on this box).  A prompt is
therefore token ids, plus — for every image — its embedding rows and grid: `{"image_embeds": [f32 [h*w, hidden], ...],
"image_grids": [(h, w), ...]}` where h x w is the MERGED token grid of the image; the placeholder tokens
`image_token_id` in the ids mark where the rows go (one placeholder per row, as the processor expands them).
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


def bench_generate(n_images: int = 5, answer_tokens: int = 64, queries: int = 2, device: int = 0) -> dict:
    """EVisRAG-7B-shaped generation (BASELINE config 5: the top retrieved pages go to the generator, one query at a
    time like src/evisrag/predict.py:128-149): random weights of the Qwen2.5-VL-7B language model, image tokens as
    precomputed embedding rows (16 x 16 merged tokens per 448 x 448 page).  Returns prefill / decode timings, the
    queries/s for this answer length and the decode step's weight-streaming rate against the HBM."""
    import time
    import torch
    cfg = GenConfig()
    t0 = time.time()
    llm = LLM(cfg, limit_mm_per_prompt={"image": max(5, n_images)}, max_model_len=4096, max_prefill=2048, device=device)
    llm.load_weights(iter_synth_gen_weights(cfg, 0, device=f"cuda:{device}", bf16=True))
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
    llm.generate([{"prompt_token_ids": ids, "multi_modal_data": {"image_embeds": embs, "image_grids": [grid] * n_images}}],
                 SamplingParams(temperature=0.1, repetition_penalty=1.05, max_tokens=4, stop_token_ids=()))      # warm-up
    pre, dec, tot = [], [], []
    for _ in range(queries):
        torch.cuda.synchronize(device); a = time.perf_counter()
        pos3 = llm.prefill(ids, embs, [grid] * n_images)
        tok = llm.sample(sp, 0)
        torch.cuda.synchronize(device); b = time.perf_counter()
        nxt = int(pos3.max()) + 1
        for step in range(1, answer_tokens):
            llm.decode(tok, nxt); nxt += 1
            tok = llm.sample(sp, step)
        torch.cuda.synchronize(device); c = time.perf_counter()
        pre.append(b - a); dec.append((c - b) / max(1, answer_tokens - 1)); tot.append(c - a)
    llm.close()
    T, p_s, d_s = len(ids), float(np.median(pre)), float(np.median(dec))
    return {
        "workload": f"Qwen2.5-VL-7B-shaped language model, bf16, random weights; prompt {T} tokens ({n_images} pages x 256 image tokens "
                    f"as embedding rows + text), {answer_tokens} answer tokens, temperature 0.1, repetition_penalty 1.05, one query "
                    "at a time; vision tower not included",
        "params_billion": round(params / 1e9, 3), "load_s": round(t_load, 1),
        "prefill_ms": round(p_s * 1e3, 2), "prefill_tokens_per_s": round(T / p_s), "prefill_tflops": round(2.0 * stream * T / p_s / 1e12, 1),
        "decode_ms_per_token": round(d_s * 1e3, 3), "decode_tokens_per_s": round(1.0 / d_s, 1),
        "queries_per_s": round(1.0 / float(np.median(tot)), 3),
        "queries_per_s_at_2048_tokens": round(1.0 / (p_s + 2047 * d_s), 4),
        "roofline": {"bound": "hbm", "kernel": "decode step: vr::gemm_skinny_kernel (M = 1 weight streaming) + attention + norms",
                     "achieved": round(stream * 2 / d_s / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(stream * 2 / d_s / 8e12, 4),
                     "bytes_per_token": stream * 2}}


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
                 limit_mm_per_prompt: Optional[Dict[str, int]] = None, max_model_len: int = 8192, max_prefill: int = 4096,
                 device: int = 0, weights=None, detokenize: Optional[Callable[[List[int]], str]] = None):
        if tensor_parallel_size != 1:
            raise ValueError("one GPU per model (predict.py:114 uses tensor_parallel_size=1)")
        if dtype not in ("bfloat16", "bf16"):
            raise ValueError("the generator computes in bf16 (predict.py:115)")
        if not isinstance(model, GenConfig):
            raise TypeError("model: a GenConfig (no checkpoint reader for the generator yet); pass weights=iterable of (name, tensor)")
        self.cfg: GenConfig = model
        self.max_images = (limit_mm_per_prompt or {"image": 5}).get("image", 5)
        self.device = int(device)
        self.detokenize = detokenize
        self.max_model_len, self.max_prefill = int(max_model_len), int(min(max_prefill, max_model_len))
        self._lib = _lib.load()
        c = self.cfg
        vc = _lib.VGConfig(c.hidden_size, c.num_hidden_layers, c.num_attention_heads, c.num_key_value_heads, c.intermediate_size,
                           c.vocab_size, self.max_model_len, self.max_prefill, c.rms_norm_eps, c.rope_theta,
                           (C.c_int32 * 3)(*c.mrope_section))
        h = C.c_void_p()
        _lib.check(self._lib.vg_create(self.device, C.byref(vc), C.byref(h)), "vg_create")
        self._h = h
        if weights is not None:
            self.load_weights(weights)

    def load_weights(self, weights) -> None:
        """weights: iterable of (HF state-dict key, torch tensor f32/bf16 on the CPU or on this GPU)."""
        import torch
        for name, t in (weights.items() if hasattr(weights, "items") else weights):
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

    def decode(self, token: int, position: int) -> None:
        p = (C.c_int32 * 3)(position, position, position)
        _lib.check(self._lib.vg_decode(self._h, int(token), p, None), "vg_decode")

    # ---- predict.py:147 ------------------------------------------------------------------------------------
    def generate(self, prompts, sampling_params: Optional[SamplingParams] = None) -> List[RequestOutput]:
        sp = sampling_params or SamplingParams()
        stops = set(sp.stop_token_ids if sp.stop_token_ids is not None else self.cfg.eos_token_ids)
        outs = []
        for pr in prompts:
            ids = list(pr["prompt_token_ids"])
            mm = pr.get("multi_modal_data") or {}
            pos3 = self.prefill(ids, mm.get("image_embeds", ()), mm.get("image_grids", ()), pr.get("positions"))
            nxt = int(pos3.max()) + 1
            toks: List[int] = []
            room = self.max_model_len - len(ids)
            for step in range(min(sp.max_tokens, room)):
                tok = self.sample(sp, step)
                toks.append(tok)
                if tok in stops or step + 1 == min(sp.max_tokens, room):
                    break
                self.decode(tok, nxt)
                nxt += 1
            text = self.detokenize(toks) if self.detokenize else ""
            outs.append(RequestOutput([CompletionOutput(toks, text)], ids))
        return outs
