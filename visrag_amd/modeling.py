"""Drop-in for the reference's retrieval-model wrapper on the encode path.

Mirrors src/openmatch/modeling/dense_retrieval_model.py:
  * `DROutput` (q_reps / p_reps)                                   :30-36
  * `DRModelForInference.build(model_args, ...)`                   :233-364, 387-391
  * `forward(query=..., passage=..., **kwargs)` / `encode_query` /
    `encode_passage`                                               :142-231, 393-408
where `query` / `passage` are the batch dicts of inference.py:85-101 (`id`, `text`, `image`
lists; extra keys are ignored) and kwargs carry `tokenizer` and `max_inp_length`.
Pooling `wmean` + `normalize=True` (the published VisRAG-Ret setting, eval.sh:62-63) are
fused into the device path; other poolings raise.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch

from .config import VisRAGRetConfig, full_config
from .engine import HipEncoder, overlapping_streams
from .preprocess import PreparedItem, prepare_batch


@dataclass
class DROutput:
    q_reps: Optional[torch.Tensor] = None
    p_reps: Optional[torch.Tensor] = None
    loss: Optional[torch.Tensor] = None
    scores: Optional[torch.Tensor] = None


class DRModelForInference:
    """HIP-backed equivalent of openmatch's DRModelForInference for VisRAG-Ret."""

    def __init__(self, cfg: VisRAGRetConfig, encoder: HipEncoder, pooling: str = "wmean",
                 normalize: bool = True, gpu_preprocess: bool = True):
        if pooling in ("drop_wmean", "drop_mean", "lasttoken_simcse"):
            # the reference applies dropout in TRAINING mode inside these even at inference (a fresh nn.Dropout1d(0.3),
            # dense_retrieval_model.py:187,196; F.dropout(training=True), :215): its own output is random
            raise ValueError(f"pooling {pooling!r} is stochastic in the reference (training-mode dropout); use "
                             "wmean / mean / lasttoken / cls")
        if pooling not in HipEncoder.POOLINGS:
            raise ValueError("Unknown pooling type: {}".format(pooling))      # dense_retrieval_model.py:220
        assert normalize == True, "Normalize must be true"   # dense_retrieval_model.py:222
        encoder.set_pooling(pooling)
        self.cfg, self.encoder = cfg, encoder
        self.pooling, self.normalize = pooling, normalize
        self.micro_batch = encoder.max_seqs
        # batches in flight: slot j = its own HipEncoder (clone: same weights, own workspace) + HIP stream.
        # Consecutive model(...) calls rotate over the slots, so batch i+1 (pre-processing included) runs
        # while batch i is still on the GPU; the caller's stream waits for each result, never the reverse.
        self._slots = [(encoder, None)]
        self._rr = 0
        self._uploaders = {}            # per slot: pinned staging + device buffer for a batch's page pixels
        # page resize + slicing on the GPU (bit-identical to PIL, gpu_resize.py) instead of on the host
        self.gpu_preprocess = gpu_preprocess

    # ---- construction -----------------------------------------------------------------------
    @classmethod
    def build(cls, model_args=None, cfg: Optional[VisRAGRetConfig] = None,
              state_dict: Optional[Iterable[Tuple[str, torch.Tensor]]] = None, device: Optional[int] = None,
              max_images: int = 32, max_tokens: int = 4096, max_seqs: int = 64, pipeline: int = 2, **_):
        """`model_args` needs `.model_name_or_path` (a HF checkpoint dir with *.safetensors /
        pytorch_model*.bin and config.json) unless `state_dict` is given; `.pooling` and
        `.normalize` are honoured like the reference (arguments.py).

        `device=None` (what the unchanged driver passes, driver/eval.py:124-127) resolves like the
        reference's `encoding_args.device`: this process's LOCAL_RANK under torchrun, else torch's
        current device — one process per GPU, never "all ranks on GPU 0"."""
        device = default_device() if device is None else _device_index(device)
        pooling = (getattr(model_args, "pooling", None) if model_args is not None else None) or \
                  (getattr(cfg, "pooling", None) if cfg is not None else None) or "wmean"
        normalize = getattr(model_args, "normalize", True) if model_args is not None else True
        path = getattr(model_args, "model_name_or_path", None) if model_args is not None else None
        if cfg is None:
            cfg = config_from_checkpoint(path) if path else full_config()
        enc = HipEncoder(cfg, device=device, max_images=max_images, max_tokens=max_tokens, max_seqs=max_seqs)
        if state_dict is None:
            if not path:
                raise ValueError("need model_args.model_name_or_path or state_dict")
            state_dict = iter_checkpoint(path)
        enc.load_state_dict(state_dict)
        model = cls(cfg, enc, pooling=pooling, normalize=normalize)
        model.set_pipeline(pipeline)
        return model

    def set_pipeline(self, depth: int) -> None:
        """depth >= 2: keep that many batches in flight (one workspace + stream each, weights shared)."""
        depth = max(1, int(depth))
        enc = self.encoder
        dev = torch.device(f"cuda:{enc.device}")
        # (streams that really overlap: two pool streams can share a hardware queue, engine.py::overlapping_streams)
        self._slots = [(enc, None)] if depth == 1 else \
            [(enc if j == 0 else enc.clone(), st) for j, st in enumerate(overlapping_streams(dev, depth))]
        self._rr = 0

    def eval(self):
        return self

    def to(self, device=None, *_a, **_k):
        """The reference driver calls `model.to(encoding_args.device)` after build()
        (driver/eval.py:128-129).  Weights already live on the build device: the same device is a
        no-op, a different one is refused (silently staying put would send every rank to one GPU)."""
        if device is None or isinstance(device, torch.dtype):
            return self
        want = _device_index(device)
        if want is not None and want != self.encoder.device:
            raise RuntimeError(f"model was built on cuda:{self.encoder.device}, to({device!r}) asks for cuda:{want}: "
                               "set LOCAL_RANK / torch.cuda.set_device() before build(), or pass build(device=...)")
        return self

    @property
    def device(self) -> torch.device:
        return torch.device(f"cuda:{self.encoder.device}")

    @property
    def lm_q(self) -> "VisRAGRet":
        """The underlying model with its HF-style forward (dense_retrieval_model.py:67-68: `lm_q` / `lm_p`, tied)."""
        return VisRAGRet(self)

    lm_p = lm_q

    # ---- reference API ----------------------------------------------------------------------
    def encode(self, items: Optional[Dict], is_query: bool = False, tokenizer=None,
               max_inp_length: int = 2048, **_):
        if items is None:
            return None, None
        if tokenizer is None:
            raise ValueError("tokenizer is required (model(passage=batch, tokenizer=tok, ...))")
        texts, images = list(items["text"]), list(items.get("image", [None] * len(items["text"])))
        enc, side = self._slots[self._rr % len(self._slots)]
        self._rr += 1

        def run():
            if self.gpu_preprocess and any(im is not None for im in images):
                from .gpu_resize import PageUploader, prepare_item_gpu
                up = self._uploaders.setdefault(id(enc), PageUploader(enc.device))
                dev_images = up.upload(images)             # the whole batch's pixels in one asynchronous copy
                prepared = [prepare_item_gpu(t, im, tokenizer, self.cfg, max_inp_length, enc.device)[0]
                            for t, im in zip(texts, dev_images)]
            else:
                prepared = prepare_batch(texts, images, tokenizer, self.cfg, max_inp_length)
            return self.encode_prepared(prepared, enc)

        if side is None:
            return None, run()
        cur = torch.cuda.current_stream(side.device)
        with torch.cuda.stream(side):          # resize kernels + encode of this batch: all on the slot's stream
            reps = run()
            done = side.record_event()
        cur.wait_event(done)                   # consumers on the caller's stream are ordered after the batch
        reps.record_stream(cur)
        return None, reps

    def encode_prepared(self, prepared: List[PreparedItem], enc: Optional[HipEncoder] = None) -> torch.Tensor:
        """Splits into micro-batches that fit the workspace (tokens / sequences).

        The precision route is chosen PER ITEM, not per batch: the library runs a micro-batch without image slices through the
        split-precision decoder pass (fp32-class; include/visrag_hip.h: text_split_precision) and one with slices through the
        bf16 pass, so token-only items are grouped into micro-batches of their own — a query's embedding no longer depends
        on whether a page happens to sit in the same call.  Item order of the result = item order of the call."""
        enc = enc or self.encoder
        text_only = [i for i, it in enumerate(prepared) if not it.slices]
        if 0 < len(text_only) < len(prepared) and getattr(self.cfg, "text_split_precision", 1):
            with_img = [i for i, it in enumerate(prepared) if it.slices]
            parts = [(text_only, self._encode_group([prepared[i] for i in text_only], enc)),
                     (with_img, self._encode_group([prepared[i] for i in with_img], enc))]
            out = torch.empty((len(prepared), parts[0][1].shape[1]), dtype=parts[0][1].dtype, device=parts[0][1].device)
            for idx, reps in parts:
                out[torch.as_tensor(idx, device=out.device)] = reps
            return out
        return self._encode_group(prepared, enc)

    def _encode_group(self, prepared: List[PreparedItem], enc: HipEncoder) -> torch.Tensor:
        outs, cur, tok = [], [], 0
        for it in prepared:
            n = len(it.input_ids)
            if n > enc.max_tokens:
                # (items are already truncated to max_inp_length like the reference's _convert_to_tensors,
                # modeling_minicpmv.py:179-180: this is the WORKSPACE being smaller than max_inp_length)
                raise ValueError(f"sequence of {n} tokens exceeds the encoder workspace (max_tokens={enc.max_tokens}): build "
                                 "with max_tokens >= max_inp_length")
            if cur and (tok + n > enc.max_tokens or len(cur) >= enc.max_seqs):
                outs.append(enc.encode_items(cur)); cur, tok = [], 0
            cur.append(it); tok += n
        if cur:
            outs.append(enc.encode_items(cur))
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    def encode_passage(self, psg, **kwargs):
        return self.encode(psg, is_query=False, **kwargs)

    def encode_query(self, qry, **kwargs):
        return self.encode(qry, is_query=True, **kwargs)

    def forward(self, query: Optional[Dict] = None, passage: Optional[Dict] = None, **kwargs) -> DROutput:
        _, q_reps = self.encode_query(query, **kwargs)
        _, p_reps = self.encode_passage(passage, **kwargs)
        return DROutput(q_reps=q_reps, p_reps=p_reps)

    __call__ = forward


@dataclass
class BaseModelOutputWithAttentionMask:
    """modeling_visrag_ret.py:24-27"""
    last_hidden_state: Optional[torch.Tensor] = None
    attention_mask: Optional[torch.Tensor] = None


class VisRAGRet:
    """The HF-style forward of the reference's SECOND caller: the demo scripts hold the bare
    `VisRAG_Ret` model and pool its output themselves,

        outputs = model(text=[...], image=[PIL | None, ...], tokenizer=tokenizer)      # demo/visrag_pipeline/utils.py:12-32
        reps = weighted_mean_pooling(outputs.last_hidden_state, outputs.attention_mask)

    (modeling_visrag_ret.py:86-126).  Same call here: `last_hidden_state` [B, L, hidden] float32 on the device, right-padded
    with zeros, `attention_mask` [B, L] int8 (modeling_minicpmv.py:464), both from ONE pass of the library
    (vr_encode_hidden).  `DRModelForInference.lm_q` / `.lm_p` are this object, like the reference's attributes."""

    def __init__(self, owner: "DRModelForInference"):
        self._owner = owner
        self.config = owner.cfg

    @classmethod
    def from_pretrained(cls, path: str, **kw) -> "VisRAGRet":
        return DRModelForInference.build(types_namespace(model_name_or_path=path), **kw).lm_q

    def eval(self):
        return self

    def to(self, *a, **k):
        self._owner.to(*a, **k)
        return self

    @property
    def device(self) -> torch.device:
        return self._owner.device

    def forward(self, text: List[str], image: List, tokenizer, max_inp_length: int = 2048, **_) -> BaseModelOutputWithAttentionMask:
        if len(text) != len(image):
            raise ValueError("text and image lists must have the same length")
        own = self._owner
        enc = own.encoder
        prepared = prepare_batch(list(text), list(image), tokenizer, own.cfg, max_inp_length)
        L = max(len(it.input_ids) for it in prepared)
        dev = torch.device(f"cuda:{enc.device}")
        hidden = torch.empty((len(prepared), L, own.cfg.hidden_size), dtype=torch.float32, device=dev)
        mask = torch.zeros((len(prepared), L), dtype=torch.int8)
        for i, it in enumerate(prepared):
            mask[i, :len(it.input_ids)] = 1
        # one library call per precision route and workspace-sized micro-batch, each writing its rows of `hidden`
        text_only = [i for i, it in enumerate(prepared) if not it.slices]
        with_img = [i for i, it in enumerate(prepared) if it.slices]
        split = bool(text_only) and bool(with_img) and bool(getattr(own.cfg, "text_split_precision", 1))
        for group in ([text_only, with_img] if split else [list(range(len(prepared)))]):
            cur, tok = [], 0
            for i in group + [None]:
                n = 0 if i is None else len(prepared[i].input_ids)
                if cur and (i is None or tok + n > enc.max_tokens or len(cur) >= enc.max_seqs):
                    contiguous = cur == list(range(cur[0], cur[0] + len(cur)))
                    part = hidden[cur[0]:cur[0] + len(cur)] if contiguous else torch.empty((len(cur), L, hidden.shape[2]), dtype=torch.float32, device=dev)
                    enc.encode_items([prepared[j] for j in cur], hidden_out=part)
                    if not contiguous:
                        hidden[torch.as_tensor(cur, device=dev)] = part
                    cur, tok = [], 0
                if i is not None:
                    if n > enc.max_tokens:
                        raise ValueError(f"sequence of {n} tokens exceeds the encoder workspace (max_tokens={enc.max_tokens})")
                    cur.append(i); tok += n
        return BaseModelOutputWithAttentionMask(last_hidden_state=hidden, attention_mask=mask.to(dev))

    __call__ = forward


def types_namespace(**kw):
    import types
    return types.SimpleNamespace(**kw)


def default_device() -> int:
    """cuda index of this process: LOCAL_RANK (torchrun / torch.distributed.run) if set, else
    torch's current device."""
    lr = os.environ.get("LOCAL_RANK")
    if lr is not None and lr.strip() != "":
        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        return int(lr) % n if n else int(lr)
    return int(torch.cuda.current_device()) if torch.cuda.is_available() else 0


def _device_index(device) -> Optional[int]:
    """int | 'cuda' | 'cuda:3' | torch.device -> cuda index ('cuda' = this process's default); None for
    anything that is not a cuda device spec (e.g. a dtype passed to .to())."""
    if isinstance(device, bool):
        return None
    if isinstance(device, int):
        return device
    if isinstance(device, str):
        try:
            device = torch.device(device)
        except (RuntimeError, ValueError):
            return None
    if isinstance(device, torch.device):
        if device.type != "cuda":
            raise RuntimeError(f"visrag_amd runs on HIP devices only (got {device}); there is no CPU fallback")
        return default_device() if device.index is None else int(device.index)
    return None


@torch.no_grad()
def encode(model: DRModelForInference, tokenizer, text_or_image_list) -> np.ndarray:
    """Demo helper with the signature of visrag_scripts/demo/visrag_pipeline/utils.py:12-32."""
    if isinstance(text_or_image_list[0], str):
        batch = {"text": list(text_or_image_list), "image": [None] * len(text_or_image_list)}
        out = model(query=batch, tokenizer=tokenizer).q_reps
    else:
        batch = {"text": [""] * len(text_or_image_list), "image": list(text_or_image_list)}
        out = model(passage=batch, tokenizer=tokenizer).p_reps
    return out.detach().cpu().numpy()


# ------------------------------------------------------------------ checkpoint plumbing ----
def config_from_checkpoint(path: str) -> VisRAGRetConfig:
    """config.json of a MiniCPM-V-2.0 / VisRAG-Ret checkpoint -> VisRAGRetConfig
    (dense_retrieval_model.py:248-269 dispatches on the same file)."""
    with open(os.path.join(path, "config.json")) as f:
        j = json.load(f)
    c = full_config()
    c.hidden_size = j.get("hidden_size", c.hidden_size)
    c.num_layers = j.get("num_hidden_layers", c.num_layers)
    c.num_heads = j.get("num_attention_heads", c.num_heads)
    c.intermediate_size = j.get("intermediate_size", c.intermediate_size)
    c.vocab_size = j.get("vocab_size", c.vocab_size)
    c.rms_norm_eps = j.get("rms_norm_eps", c.rms_norm_eps)
    c.rope_theta = j.get("rope_theta", c.rope_theta)
    c.scale_emb = j.get("scale_emb", c.scale_emb)
    c.scale_depth = j.get("scale_depth", c.scale_depth)
    c.query_num = j.get("query_num", c.query_num)
    c.patch_size = j.get("patch_size", c.patch_size)
    c.max_slice_nums = j.get("max_slice_nums", c.max_slice_nums)
    c.scale_resolution = j.get("scale_resolution", c.scale_resolution)
    c.slice_mode = j.get("slice_mode", c.slice_mode)
    return c


def iter_checkpoint(path: str):
    """Yield (key, tensor) from *.safetensors (preferred) or pytorch_model*.bin shards."""
    files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
    if files:
        from safetensors import safe_open
        for fn in files:
            with safe_open(os.path.join(path, fn), framework="pt", device="cpu") as f:
                for k in f.keys():
                    yield k, f.get_tensor(k)
        return
    files = sorted(f for f in os.listdir(path) if f.startswith("pytorch_model") and f.endswith(".bin"))
    if not files:
        raise FileNotFoundError(f"no *.safetensors / pytorch_model*.bin under {path}")
    for fn in files:
        sd = torch.load(os.path.join(path, fn), map_location="cpu", weights_only=True)
        for k, v in sd.items():
            yield k, v
