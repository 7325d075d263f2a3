"""ctypes binding of libvisrag_hip.so (include/visrag_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails the
product raises — a silent CPU/PyTorch path would void every parity and performance claim.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvisrag_hip.so")

VR_DTYPE_F32, VR_DTYPE_BF16 = 0, 1


class VRConfig(C.Structure):
    _fields_ = [
        ("patch_size", C.c_int32), ("vit_dim", C.c_int32), ("vit_depth", C.c_int32),
        ("vit_heads", C.c_int32), ("vit_hidden", C.c_int32), ("vit_pos_grid", C.c_int32),
        ("vit_ln_eps", C.c_float), ("query_num", C.c_int32), ("resampler_ln_eps", C.c_float),
        ("hidden_size", C.c_int32), ("num_layers", C.c_int32), ("num_heads", C.c_int32),
        ("intermediate_size", C.c_int32), ("vocab_size", C.c_int32), ("rms_norm_eps", C.c_float),
        ("rope_theta", C.c_float), ("scale_emb", C.c_float), ("residual_scale", C.c_float),
        ("max_images", C.c_int32), ("max_patches", C.c_int32), ("max_tokens", C.c_int32),
        ("max_seqs", C.c_int32), ("text_split_precision", C.c_int32),
    ]


class VGConfig(C.Structure):       # vg_config_t (include/visrag_gen.h)
    _fields_ = [
        ("hidden_size", C.c_int32), ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32),
        ("intermediate_size", C.c_int32), ("vocab_size", C.c_int32), ("max_len", C.c_int32), ("max_prefill", C.c_int32),
        ("rms_norm_eps", C.c_float), ("rope_theta", C.c_float), ("mrope_section", C.c_int32 * 3), ("max_seqs", C.c_int32),
    ]


class VGVisionConfig(C.Structure):     # vg_vision_config_t (include/visrag_gen.h)
    _fields_ = [
        ("depth", C.c_int32), ("hidden_size", C.c_int32), ("num_heads", C.c_int32), ("intermediate_size", C.c_int32),
        ("out_hidden_size", C.c_int32), ("in_channels", C.c_int32), ("patch_size", C.c_int32), ("temporal_patch_size", C.c_int32),
        ("spatial_merge_size", C.c_int32), ("window_size", C.c_int32), ("n_fullatt", C.c_int32), ("fullatt_blocks", C.c_int32 * 16),
        ("max_rows", C.c_int32), ("rms_norm_eps", C.c_float),
    ]


class VisragHipError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None

# name -> (restype, argtypes); every symbol include/visrag_hip.h declares
_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
SIGNATURES = {
    "vr_version": (C.c_char_p, []),
    "vr_last_error": (C.c_char_p, []),
    "vr_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "vr_model_create": (C.c_int, [C.c_int, C.POINTER(VRConfig), C.POINTER(_vp)]),
    "vr_model_destroy": (C.c_int, [_vp]),
    "vr_model_load_weight": (C.c_int, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i32, _i32, _i32]),
    "vr_model_finalize": (C.c_int, [_vp]),
    "vr_model_clone": (C.c_int, [_vp, C.POINTER(C.c_void_p)]),
    "vr_encode": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_i32), _i32, _i32, C.POINTER(_i32),
                            C.POINTER(_i32), _i32, C.POINTER(_i32), _vp, _i32, _vp]),
    "vr_encode_hidden": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_i32), _i32, _i32, C.POINTER(_i32),
                                   C.POINTER(_i32), _i32, C.POINTER(_i32), _vp, _i32, _vp, _i32, _vp]),
    "vr_model_tap": (C.c_int, [_vp, C.c_char_p, _vp, _i64, _i64]),
    "vr_model_set_taps": (C.c_int, [_vp, _i32]),
    "vr_model_set_pooling": (C.c_int, [_vp, _i32]),
    "vr_model_set_profile": (C.c_int, [_vp, _i32]),
    "vr_model_get_profile": (C.c_int, [_vp, _i32, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double)]),
    "vr_index_create": (C.c_int, [C.c_int, _i32, _i64, C.POINTER(_vp)]),
    "vr_index_destroy": (C.c_int, [_vp]),
    "vr_index_reset": (C.c_int, [_vp]),
    "vr_index_add": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "vr_index_size": (C.c_int, [_vp, C.POINTER(_i64)]),
    "vr_index_search": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp]),
    "vr_topk_merge": (C.c_int, [C.c_int, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "vr_index_set_search_eps": (C.c_int, [_vp, _f32]),
    "vr_index_search_stats": (C.c_int, [_vp, C.POINTER(_i64), _i32]),
    "vr_index_error_model": (C.c_int, [_vp, C.POINTER(_f32)]),
    "vr_index_search_plan": (C.c_int, [_vp, _i32, C.POINTER(_i32)]),
    "vr_synth_pages": (C.c_int, [C.c_int, _vp, _i32, _i32, _i64, _i64, _vp]),
    "vr_streams_overlap": (C.c_int, [C.c_int, _vp, _vp, C.POINTER(C.c_int32)]),
    "vr_index_set_search_profile": (C.c_int, [_vp, _i32]),
    "vr_index_get_search_profile": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "vr_index_search_keys": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _vp, _i32, _vp]),
    "vr_topk_merge_keys": (C.c_int, [C.c_int, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "vr_resize_bicubic": (C.c_int, [C.c_int, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "vr_op_gemm": (C.c_int, [C.c_int, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _f32,
                             _vp, _i32, _vp, _vp, _i32, _i32, _vp]),
    "vr_op_norm": (C.c_int, [C.c_int, _i32, _vp, _i32, _i32, _vp, _vp, _f32, _vp, _i32, _vp]),
    "vr_op_attention": (C.c_int, [C.c_int, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _i32,
                                  _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
}
# every symbol include/visrag_gen.h declares (the EVisRAG generator's language model)
GEN_SIGNATURES = {
    "vg_create": (C.c_int, [C.c_int, C.POINTER(VGConfig), C.POINTER(_vp)]),
    "vg_destroy": (C.c_int, [_vp]),
    "vg_load_weight": (C.c_int, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i32, _i32, _i32]),
    "vg_finalize": (C.c_int, [_vp]),
    "vg_prefill": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp]),
    "vg_sample": (C.c_int, [_vp, _f32, _f32, C.c_uint64, _i32, C.POINTER(_i32), _vp]),
    "vg_decode": (C.c_int, [_vp, _i32, C.POINTER(_i32), _vp]),
    "vg_logits": (C.c_int, [_vp, _vp, _vp]),
    "vg_select": (C.c_int, [_vp, _i32]),
    "vg_decode_batch": (C.c_int, [_vp, _i32, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), _vp]),
    "vg_sample_batch": (C.c_int, [_vp, _i32, C.POINTER(_i32), _f32, _f32, C.c_uint64, _i32, C.POINTER(_i32), _vp]),
    "vg_cache_len": (C.c_int, [_vp, C.POINTER(_i32)]),
    "vg_run_begin": (C.c_int, [_vp, _i32, _f32, _f32, C.c_uint64, _i32, _vp]),
    "vg_run_step": (C.c_int, [_vp]),
    "vg_run_token": (C.c_int, [_vp, _i32, C.POINTER(_i32)]),
    "vg_run_end": (C.c_int, [_vp]),
    "vg_vision_create": (C.c_int, [_vp, C.POINTER(VGVisionConfig)]),
    "vg_vision_encode": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp]),
    "vg_vision_encode_pages": (C.c_int, [_vp, C.POINTER(_vp), _i32, C.POINTER(_f32), C.POINTER(_f32), _vp, _i32, _vp, _vp]),
    "vg_vision_plan": (C.c_int, [C.POINTER(VGVisionConfig), _vp, _i32, _vp, _vp, C.POINTER(_i32), _vp]),
}


def load(path: Optional[str] = None) -> C.CDLL:
    """dlopen the library and bind every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # VISRAG_HIP_LIB: load another build of the SAME library (tools/ab_*.py: tagged A/B builds of
    # `python -m visrag_amd.build --tag ...`); it is not a fallback — a missing file still raises
    p = path or os.environ.get("VISRAG_HIP_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise VisragHipError(
            f"{p} not found: the HIP extension is not built. Run `python -m visrag_amd.build` "
            "(or __graft_entry__.build()); there is no CPU fallback.")
    # torch wheels bundle their own ROCm runtime (torch/lib/libamdhip64.so).  It must be the
    # ONE HIP runtime of the process: import torch first so that our DT_NEEDED libamdhip64.so.7
    # binds to the already-loaded copy (loading ours first would pull /opt/rocm's runtime in and
    # give torch a second one — streams and device pointers would not be shared).
    import torch  # noqa: F401
    lib = C.CDLL(p)
    for name, (res, args) in list(SIGNATURES.items()) + list(GEN_SIGNATURES.items()):
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().vr_last_error()
        raise VisragHipError(f"{what or 'visrag_hip'} failed (status {status}): "
                             f"{msg.decode() if msg else '?'}")


def make_config(cfg, max_images: int, max_patches: int, max_tokens: int, max_seqs: int) -> VRConfig:
    return VRConfig(
        patch_size=cfg.patch_size, vit_dim=cfg.vit_dim, vit_depth=cfg.vit_depth, vit_heads=cfg.vit_heads,
        vit_hidden=cfg.vit_hidden, vit_pos_grid=cfg.vit_pos_grid, vit_ln_eps=cfg.vit_ln_eps,
        query_num=cfg.query_num, resampler_ln_eps=cfg.resampler_ln_eps, hidden_size=cfg.hidden_size,
        num_layers=cfg.num_layers, num_heads=cfg.num_heads, intermediate_size=cfg.intermediate_size,
        vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
        scale_emb=cfg.scale_emb, residual_scale=cfg.residual_scale, max_images=max_images,
        max_patches=max_patches, max_tokens=max_tokens, max_seqs=max_seqs,
        text_split_precision=int(getattr(cfg, "text_split_precision", True)))
