"""Thin Python owners of the two C-ABI objects: the VisRAG-Ret encoder (`HipEncoder`) and the
HBM-resident index (`HipIndex`).  torch tensors are used only as device-memory containers
and for the current HIP stream; all arithmetic happens in libvisrag_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .config import VisRAGRetConfig
from .preprocess import PreparedItem


def _stream_ptr(device: int = 0) -> int:
    """HIP stream handle of torch's CURRENT stream ON `device` (the device the C side launches on)."""
    return int(torch.cuda.current_stream(int(device)).cuda_stream) if torch.cuda.is_available() else 0


def _require_gpu():
    if not torch.cuda.is_available():
        raise _lib.VisragHipError("no HIP device visible: visrag_amd has no CPU fallback")


def streams_overlap(a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> bool:
    """Do kernels on the two streams run side by side (vr_streams_overlap: two spin kernels), or did the runtime put the
    streams on one hardware queue?"""
    _require_gpu()
    got = C.c_int32(0)
    _lib.check(_lib.load().vr_streams_overlap(int(a.device.index), C.c_void_p(int(a.cuda_stream)), C.c_void_p(int(b.cuda_stream)),
                                              C.byref(got)), "vr_streams_overlap")
    return bool(got.value)


def overlapping_streams(device, n: int = 2, tries: int = 12) -> List["torch.cuda.Stream"]:
    """`n` torch streams on `device` whose kernels overlap pairwise.  torch hands its pool streams out round-robin and
    HIP maps them onto four hardware queues: two NEIGHBOURING pool streams can share a queue (streams 2 and 3 of a fresh
    process do), and two batches "in flight" on them run at the single-stream rate — 661 against 711 pages/s.  Streams are
    drawn from the pool until `n` are found that overlap pairwise (probed, ~2.5 ms a pair); if the pool cannot deliver
    within `tries` draws the last draws fill the list — slower, still correct."""
    dev = torch.device(device) if not isinstance(device, int) else torch.device(f"cuda:{device}")
    chosen: List[torch.cuda.Stream] = []
    spare: List[torch.cuda.Stream] = []
    for _ in range(max(tries, n)):
        s = torch.cuda.Stream(device=dev)
        if all(streams_overlap(s, c) for c in chosen):
            chosen.append(s)
            if len(chosen) == n:
                return chosen
        else:
            spare.append(s)
    return (chosen + spare)[:n]


class HipEncoder:
    """Device-resident VisRAG-Ret weights + workspace (vr_model_*)."""

    def __init__(self, cfg: VisRAGRetConfig, device: int = 0, max_images: int = 32,
                 max_patches: Optional[int] = None, max_tokens: int = 4096, max_seqs: int = 64):
        _require_gpu()
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = int(device)
        if max_patches is None:
            g = cfg.scale_resolution // cfg.patch_size
            max_patches = int(g * g * 1.1) + 8       # sliced pages reach ~1064 patches per slice
        self.max_images, self.max_tokens, self.max_seqs = max_images, max_tokens, max_seqs
        self._c = _lib.make_config(cfg, max_images, max_patches, max_tokens, max_seqs)
        self._h = C.c_void_p()
        _lib.check(self.lib.vr_model_create(self.device, C.byref(self._c), C.byref(self._h)), "vr_model_create")
        self.finalized = False

    def clone(self) -> "HipEncoder":
        """A second encoder on the SAME device weights with its own workspace (vr_model_clone): run it on
        another torch stream to keep two batches in flight.  Keeps a reference to its source."""
        if not self.finalized:
            raise RuntimeError("clone() needs a finalized encoder")
        other = object.__new__(HipEncoder)
        other.lib, other.cfg, other.device = self.lib, self.cfg, self.device
        other.max_images, other.max_tokens, other.max_seqs = self.max_images, self.max_tokens, self.max_seqs
        other._c = self._c
        other._h = C.c_void_p()
        other._src = self
        _lib.check(self.lib.vr_model_clone(self._h, C.byref(other._h)), "vr_model_clone")
        other.finalized = True
        return other

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.vr_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    POOLINGS = {"wmean": 0, "mean": 1, "lasttoken": 2, "cls": 3}

    def set_pooling(self, pooling: str) -> None:
        """DRModel.encode's pooling (dense_retrieval_model.py:172-220); clones made afterwards inherit it."""
        if pooling not in self.POOLINGS:
            raise ValueError("Unknown pooling type: {}".format(pooling))
        _lib.check(self.lib.vr_model_set_pooling(self._h, self.POOLINGS[pooling]), "vr_model_set_pooling")

    # ---- weights --------------------------------------------------------------------------
    def load_weight(self, name: str, t: torch.Tensor) -> None:
        if t.dtype == torch.bfloat16:
            dt = _lib.VR_DTYPE_BF16
        else:
            t = t.to(torch.float32)
            dt = _lib.VR_DTYPE_F32
        t = t.contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        _lib.check(self.lib.vr_model_load_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), shape,
                                                 t.dim(), dt, 1 if t.is_cuda else 0), f"load {name}")
        self.finalized = False

    def load_state_dict(self, items: Iterable[Tuple[str, torch.Tensor]]) -> None:
        """items: (HF key, tensor) pairs — a dict's .items() or a generator (tensors may be
        released by the caller as soon as the call returns)."""
        if isinstance(items, dict):
            items = items.items()
        for name, t in items:
            self.load_weight(name, t)
        self.finalize()

    def finalize(self) -> None:
        _lib.check(self.lib.vr_model_finalize(self._h), "vr_model_finalize")
        self.finalized = True

    # ---- encode ---------------------------------------------------------------------------
    def encode_items(self, items: Sequence[PreparedItem], device_slices: Optional[List[torch.Tensor]] = None,
                     out: Optional[torch.Tensor] = None, hidden_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """items -> unit-norm embeddings [B, hidden] float32 on the device.
        `device_slices`: optional uint8 HWC cuda tensors replacing the host slices of the items
        (same order: item 0's slices first, ...).
        `hidden_out`: optional float32 cuda tensor [B, L, hidden] (L >= the longest item) that receives the last hidden
        states, right-padded with zeros (vr_encode_hidden: the HF-style forward's `last_hidden_state`)."""
        cfg = self.cfg
        B = len(items)
        if B == 0:
            raise ValueError("empty batch")
        Q = cfg.query_num
        seq = np.zeros(B + 1, dtype=np.int32)
        for i, it in enumerate(items):
            seq[i + 1] = seq[i] + len(it.input_ids)
        ids = np.concatenate([np.asarray(it.input_ids, dtype=np.int32) for it in items])
        slices: List = []
        hw: List[int] = []
        rows: List[np.ndarray] = []
        di = 0
        for i, it in enumerate(items):
            for k, s in enumerate(it.slices):
                if device_slices is not None:     # shapes come from the device tensors (host slice may be None)
                    s = device_slices[di]
                di += 1
                slices.append(s)
                hw += [int(s.shape[0]), int(s.shape[1])]
                r = np.full(Q, -1, dtype=np.int32)
                if k < len(it.image_bound):       # bound k <-> slice k (scatter_ semantics)
                    b0, b1 = it.image_bound[k]
                    n = max(0, min(Q, b1 - b0))
                    r[:n] = seq[i] + b0 + np.arange(n, dtype=np.int32)
                rows.append(r)
        n_slices = len(slices)
        if device_slices is None and n_slices and all(isinstance(t, torch.Tensor) and t.is_cuda for t in slices):
            device_slices = slices            # items prepared on the GPU (gpu_resize.prepare_item_gpu)
        on_dev = 0
        keep = []
        if n_slices:
            if device_slices is not None:
                if len(device_slices) != n_slices:
                    raise ValueError("device_slices must match the items' slices")
                ptrs = (C.c_void_p * n_slices)(*[C.c_void_p(t.data_ptr()) for t in device_slices])
                on_dev = 1
            else:
                keep = [np.ascontiguousarray(s, dtype=np.uint8) for s in slices]
                ptrs = (C.c_void_p * n_slices)(*[C.c_void_p(a.ctypes.data) for a in keep])
            hw_arr = (C.c_int32 * (2 * n_slices))(*hw)
            vr = np.ascontiguousarray(np.concatenate(rows), dtype=np.int32)
            vr_p = vr.ctypes.data_as(C.POINTER(C.c_int32))
        else:
            ptrs, hw_arr, vr_p = None, None, None
        if out is None:
            out = torch.empty((B, cfg.hidden_size), dtype=torch.float32, device=f"cuda:{self.device}")
        if hidden_out is not None:
            if not (hidden_out.is_cuda and hidden_out.dtype == torch.float32 and hidden_out.is_contiguous() and hidden_out.dim() == 3
                    and hidden_out.shape[0] == B and hidden_out.shape[2] == cfg.hidden_size):
                raise ValueError("hidden_out must be a contiguous float32 cuda tensor [B, L, hidden_size]")
            _lib.check(self.lib.vr_encode_hidden(
                self._h, ptrs, hw_arr, n_slices, on_dev,
                ids.ctypes.data_as(C.POINTER(C.c_int32)), seq.ctypes.data_as(C.POINTER(C.c_int32)), B,
                vr_p, C.c_void_p(out.data_ptr()), 1, C.c_void_p(hidden_out.data_ptr()), int(hidden_out.shape[1]),
                C.c_void_p(_stream_ptr(self.device))), "vr_encode_hidden")
            return out
        _lib.check(self.lib.vr_encode(
            self._h, ptrs, hw_arr, n_slices, on_dev,
            ids.ctypes.data_as(C.POINTER(C.c_int32)), seq.ctypes.data_as(C.POINTER(C.c_int32)), B,
            vr_p, C.c_void_p(out.data_ptr()), 1, C.c_void_p(_stream_ptr(self.device))), "vr_encode")
        return out

    # ---- profiling (HIP events around kernel classes, see include/visrag_hip.h) -------------
    PROF_CLASSES = ("vit_qkv", "vit_attn", "vit_proj", "vit_fc1", "vit_fc2", "resampler", "decoder",
                    "dec_qkv_rope", "dec_attn", "dec_o", "dec_gate_up", "dec_down", "dec_norms")

    def set_profile(self, on) -> None:
        """False / 0: off; True / 1: the seven phase classes; 2: the decoder's sub-phases (dec_*) instead."""
        _lib.check(self.lib.vr_model_set_profile(self._h, int(on)))

    def get_profile(self) -> Dict[str, Dict[str, float]]:
        out = {}
        for i, name in enumerate(self.PROF_CLASSES):
            ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
            _lib.check(self.lib.vr_model_get_profile(self._h, i, C.byref(ms), C.byref(n), C.byref(fl)))
            out[name] = {"ms": ms.value, "launches": int(n.value), "flops": fl.value}
        return out

    # ---- debug taps -----------------------------------------------------------------------
    def set_taps(self, on: bool) -> None:
        _lib.check(self.lib.vr_model_set_taps(self._h, 1 if on else 0))

    def tap(self, name: str, rows: int, cols: int) -> np.ndarray:
        out = np.empty((rows, cols), dtype=np.float32)
        _lib.check(self.lib.vr_model_tap(self._h, name.encode(), C.c_void_p(out.ctypes.data), rows, cols),
                   f"tap {name}")
        return out


class HipIndex:
    """HBM-resident embedding index (fp32 rows + a bf16 copy for the MFMA sweep)."""

    def __init__(self, dim: int, capacity: int, device: int = 0):
        _require_gpu()
        self.lib = _lib.load()
        self.dim, self.capacity, self.device = int(dim), int(capacity), int(device)
        self._h = C.c_void_p()
        _lib.check(self.lib.vr_index_create(self.device, self.dim, self.capacity, C.byref(self._h)),
                   "vr_index_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.vr_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        n = C.c_int64()
        _lib.check(self.lib.vr_index_size(self._h, C.byref(n)))
        return int(n.value)

    def reset(self) -> None:
        _lib.check(self.lib.vr_index_reset(self._h))

    def add(self, reps) -> None:
        if isinstance(reps, torch.Tensor):
            t = reps.to(torch.float32).contiguous()
            assert t.dim() == 2 and t.shape[1] == self.dim
            _lib.check(self.lib.vr_index_add(self._h, C.c_void_p(t.data_ptr()), t.shape[0],
                                             1 if t.is_cuda else 0, C.c_void_p(_stream_ptr(self.device))), "vr_index_add")
        else:
            a = np.ascontiguousarray(reps, dtype=np.float32)
            assert a.ndim == 2 and a.shape[1] == self.dim
            _lib.check(self.lib.vr_index_add(self._h, C.c_void_p(a.ctypes.data), a.shape[0], 0,
                                             C.c_void_p(_stream_ptr(self.device))), "vr_index_add")

    def search(self, queries, k: int):
        """-> (scores [nq,k] f32, ids [nq,k] i64); torch cuda tensors in -> cuda tensors out,
        numpy / cpu in -> numpy out."""
        if isinstance(queries, torch.Tensor) and queries.is_cuda:
            q = queries.to(torch.float32).contiguous()
            nq = q.shape[0]
            sc = torch.empty((nq, k), dtype=torch.float32, device=q.device)
            ix = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            _lib.check(self.lib.vr_index_search(self._h, C.c_void_p(q.data_ptr()), nq, k,
                                                C.c_void_p(sc.data_ptr()), C.c_void_p(ix.data_ptr()), 1,
                                                C.c_void_p(_stream_ptr(self.device))), "vr_index_search")
            return sc, ix
        q = np.ascontiguousarray(queries.numpy() if isinstance(queries, torch.Tensor) else queries,
                                 dtype=np.float32)
        nq = q.shape[0]
        sc = np.empty((nq, k), dtype=np.float32)
        ix = np.empty((nq, k), dtype=np.int64)
        _lib.check(self.lib.vr_index_search(self._h, C.c_void_p(q.ctypes.data), nq, k,
                                            C.c_void_p(sc.ctypes.data), C.c_void_p(ix.ctypes.data), 0,
                                            C.c_void_p(_stream_ptr(self.device))), "vr_index_search")
        return sc, ix


    def search_keys(self, queries: torch.Tensor, k: int, id_offset: int = 0) -> torch.Tensor:
        """The same search with each result packed into ONE 64-bit word (include/visrag_hip.h:
        vr_index_search_keys): orderable(score) << 32 | ~(row + id_offset); 0 = empty slot.  -> int64 [nq, k]
        on the queries' device — the buffer the ranks all-gather (retriever.sharded_search)."""
        q = queries.to(torch.float32).contiguous()
        if not q.is_cuda:
            raise ValueError("search_keys works on device tensors (the exchange buffer lives in HBM)")
        nq = q.shape[0]
        keys = torch.empty((nq, k), dtype=torch.int64, device=q.device)
        _lib.check(self.lib.vr_index_search_keys(self._h, C.c_void_p(q.data_ptr()), nq, k, int(id_offset),
                                                 C.c_void_p(keys.data_ptr()), 1, C.c_void_p(_stream_ptr(self.device))),
                   "vr_index_search_keys")
        return keys

    def set_search_eps(self, eps_rel: Optional[float]) -> None:
        """Error model of the top-k certification: None = the rigorous data-dependent default, >= 0 = eps_rel * |q| * max|d|,
        < 0 = certification off."""
        _lib.check(self.lib.vr_index_set_search_eps(self._h, float("nan") if eps_rel is None else float(eps_rel)))

    def search_stats(self, reset: bool = False) -> Dict[str, int]:
        """Queries since the last reset by outcome: certified from the sweep's candidates at once / after re-scoring more of
        them; `flagged` = redone behind the sweep, of which `band_pass` by re-scoring every row inside the error band and
        `exact_pass` = band beyond 8192 rows (the fp32 pass over the whole index, or — the first such queries of an index — one workgroup's walk); `uncertified` = searched with certification off."""
        out = (C.c_int64 * 6)()
        _lib.check(self.lib.vr_index_search_stats(self._h, out, 1 if reset else 0))
        return {"certified": int(out[0]), "certified_extended": int(out[1]), "flagged": int(out[2]),
                "band_pass": int(out[2]) - int(out[5]), "exact_pass": int(out[5]),
                "uncertified": int(out[3]), "regathered": int(out[4])}

    def search_plan(self, nq: int) -> Dict[str, int]:
        """How a search of `nq` queries (k <= 26) is laid out over the rows added so far (include/visrag_hip.h:
        vr_index_search_plan): `prepass_chunks` > 0 = the threshold pre-pass owns its sampled tiles, the sweep skips them."""
        out = (C.c_int32 * 5)()
        _lib.check(self.lib.vr_index_search_plan(self._h, int(nq), out))
        return {"list_chunks": int(out[0]), "prepass_chunks": int(out[1]), "sweep_chunks": int(out[2]), "tiles_per_chunk": int(out[3]),
                "exact_pass_launched": int(out[4])}

    def error_model(self) -> Dict[str, float]:
        """What the default certification bound is made of (include/visrag_hip.h: vr_index_set_search_eps)."""
        out = (C.c_float * 4)()
        _lib.check(self.lib.vr_index_error_model(self._h, out))
        return {"max_row_norm": float(out[0]), "max_row_bf16_residual": float(out[1]), "acc_rel": float(out[2]),
                "worst_case_eps_rel": float(out[3])}

    SEARCH_STAGES = ("convert", "thresholds", "sweep", "merge", "exact_pass")

    def set_search_profile(self, on: bool) -> None:
        _lib.check(self.lib.vr_index_set_search_profile(self._h, 1 if on else 0))

    def get_search_profile(self) -> Dict[str, float]:
        ms, calls = (C.c_double * 5)(), C.c_int64()
        _lib.check(self.lib.vr_index_get_search_profile(self._h, ms, C.byref(calls)))
        n = max(int(calls.value), 1)
        out = {name: ms[i] / n for i, name in enumerate(self.SEARCH_STAGES)}
        out["calls"] = int(calls.value)
        return out


def topk_merge_keys(keys: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[n_parts, nq, k] packed keys (HipIndex.search_keys of every shard, e.g. the all-gather's output as it is)
    -> merged (scores [nq, k] f32, global ids [nq, k] i64) on the device."""
    _require_gpu()
    lib = _lib.load()
    n_parts, nq, k = keys.shape
    keys = keys.contiguous()
    os_ = torch.empty((nq, k), dtype=torch.float32, device=keys.device)
    oi = torch.empty((nq, k), dtype=torch.int64, device=keys.device)
    dev = keys.device.index or 0
    _lib.check(lib.vr_topk_merge_keys(dev, C.c_void_p(keys.data_ptr()), n_parts, nq, k, C.c_void_p(os_.data_ptr()),
                                      C.c_void_p(oi.data_ptr()), C.c_void_p(_stream_ptr(dev))), "vr_topk_merge_keys")
    return os_, oi


def topk_merge(scores: torch.Tensor, ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[n_parts, nq, k] per-shard (score, global id) lists -> merged [nq, k] (device tensors)."""
    _require_gpu()
    lib = _lib.load()
    n_parts, nq, k = scores.shape
    scores = scores.contiguous().to(torch.float32)
    ids = ids.contiguous().to(torch.int64)
    os_ = torch.empty((nq, k), dtype=torch.float32, device=scores.device)
    oi = torch.empty((nq, k), dtype=torch.int64, device=scores.device)
    dev = scores.device.index or 0
    _lib.check(lib.vr_topk_merge(dev, C.c_void_p(scores.data_ptr()), C.c_void_p(ids.data_ptr()),
                                 n_parts, nq, k, C.c_void_p(os_.data_ptr()), C.c_void_p(oi.data_ptr()),
                                 C.c_void_p(_stream_ptr(dev))), "vr_topk_merge")
    return os_, oi
