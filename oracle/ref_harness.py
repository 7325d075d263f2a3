"""TEST INFRASTRUCTURE ONLY (never imported by the product path).

Runs the reference's OWN code (`/root/reference/src/openmatch`, `timm_modified`) on CPU in
this container, with the three compatibility shims SURVEY.md section 8c lists, so that
  (1) oracle/visrag_ret_oracle.py (the restatement that travels to the GPU box) can be
      pinned against the real reference, and
  (2) golden fixtures can be generated (oracle/gen_golden.py -> tests/golden/).
`/root/reference` does not exist on the GPU box; nothing under tests -m gpu / bench.py /
smoke() imports this module.

Shims (all environment-compat, none changes the arithmetic):
  * `torchvision` is not installed: a sys.meta_path stub supplies the few names the
    reference imports (timm/layers/create_norm.py:14, norm_act.py:20,
    timm/data/transforms.py:70, modeling_minicpmv.py:9-13,84-92, resampler.py:19-20).
    ToTensor/Normalize/Compose follow torchvision semantics: u8 HWC -> f32 CHW /255,
    (x-mean)/std.
  * transformers 5.x: `is_torch_fx_available` removed (modeling_minicpm.py:57,69-73).
  * transformers 5.x config: `rope_scaling=None`, `use_cache=False`
    (modeling_minicpm.py:400-409,1196-1200).
"""
from __future__ import annotations

import enum
import importlib.abc
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("VISRAG_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "openmatch"))


# ------------------------------------------------------------------ torchvision stub ----
class _InterpolationMode(enum.Enum):
    NEAREST = "nearest"
    NEAREST_EXACT = "nearest-exact"
    BILINEAR = "bilinear"
    BICUBIC = "bicubic"
    BOX = "box"
    HAMMING = "hamming"
    LANCZOS = "lanczos"


class _Compose:
    def __init__(self, transforms): self.transforms = transforms
    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class _ToTensor:
    def __call__(self, pic):
        import numpy as np, torch
        arr = np.asarray(pic)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).contiguous()
        if t.dtype == torch.uint8:
            return t.to(torch.float32).div(255)
        return t.to(torch.float32)


class _Normalize:
    def __init__(self, mean, std, inplace=False): self.mean, self.std = mean, std
    def __call__(self, t):
        import torch
        mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
        return (t - mean) / std


class _StubModule(types.ModuleType):
    __version__ = "0.0.0"
    __path__: list = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        known = {"InterpolationMode": _InterpolationMode, "Compose": _Compose,
                 "ToTensor": _ToTensor, "Normalize": _Normalize}
        if name in known:
            return known[name]
        if name == "FrozenBatchNorm2d":
            import torch
            return type("FrozenBatchNorm2d", (torch.nn.Module,), {})
        if name[:1].islower():
            # submodule access such as torchvision.transforms
            return importlib.import_module(self.__name__ + "." + name)
        return type(name, (), {})


class _TorchvisionFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == "torchvision" or fullname.startswith("torchvision."):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


_installed = False


def install_shims():
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    try:
        import torchvision  # noqa: F401  (real one present: no stub)
    except ModuleNotFoundError:
        sys.meta_path.insert(0, _TorchvisionFinder())
    import transformers.utils.import_utils as iu
    if not hasattr(iu, "is_torch_fx_available"):
        iu.is_torch_fx_available = lambda: False
    import transformers.utils as tu
    if not hasattr(tu, "is_torch_fx_available"):
        tu.is_torch_fx_available = lambda: False
    for p in (os.path.join(REFERENCE_ROOT, "timm_modified"), os.path.join(REFERENCE_ROOT, "src")):
        if p not in sys.path:
            sys.path.insert(0, p)
    _installed = True


# ------------------------------------------------------------------ model construction --
def build_reference_model(cfg, state_dict, attn_implementation: str = "sdpa"):
    """Instantiate the reference `VisRAG_Ret` (fp32, CPU, eval) with `cfg` dims and load the
    given HF-keyed state dict.  Small fixture configs inject the ViT dims through a
    `timm.create_model` wrapper (MiniCPMV.init_vision_module hard-codes the model name,
    modeling_minicpmv.py:57-73); the code path executed is unchanged."""
    install_shims()
    import torch
    import timm
    from openmatch.modeling.modeling_minicpmv.configuration_minicpm import MiniCPMVConfig
    from openmatch.modeling.modeling_visrag_ret.modeling_visrag_ret import VisRAG_Ret

    hf_cfg = MiniCPMVConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size,
        intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_layers,
        num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_heads,
        max_position_embeddings=4096, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
        scale_emb=cfg.scale_emb, dim_model_base=256, scale_depth=cfg.scale_depth,
        query_num=cfg.query_num, image_size=448, patch_size=cfg.patch_size,
        max_slice_nums=cfg.max_slice_nums, scale_resolution=cfg.scale_resolution,
        slice_mode=cfg.slice_mode, drop_vision_last_layer=True, use_cache=False,
        attn_implementation=attn_implementation,
    )
    hf_cfg.rope_scaling = None
    hf_cfg.use_cache = False
    hf_cfg._attn_implementation = attn_implementation

    orig_create = timm.create_model

    def create_model(name, **kw):
        kw.update(embed_dim=cfg.vit_dim, depth=cfg.vit_depth + 1, num_heads=cfg.vit_heads,
                  mlp_ratio=cfg.vit_mlp_ratio, img_size=cfg.vit_pos_grid * cfg.patch_size)
        return orig_create(name, **kw)

    timm.create_model = create_model
    try:
        torch.manual_seed(0)
        model = VisRAG_Ret(hf_cfg)
    finally:
        timm.create_model = orig_create
    model = model.float().eval()
    own = model.state_dict()
    missing = [k for k in state_dict if k not in own]
    assert not missing, f"synthetic keys not in the reference model: {missing[:5]}"
    res = model.load_state_dict(state_dict, strict=False)
    # keys we deliberately do not synthesise: lm_head (unused by the embedding path),
    # resampler.pos_embed (fixed sincos buffer), rotary buffers (non-persistent)
    allowed = ("llm.lm_head.", "resampler.pos_embed")
    bad = [k for k in res.missing_keys if not k.startswith(allowed)]
    assert not bad, f"reference params left uninitialised: {bad[:8]}"
    return model


def build_reference_dr_model(cfg, state_dict, pooling="wmean"):
    """The reference's DRModelForInference wrapper (dense_retrieval_model.py:387-408)
    around the model above: forward(query=..., passage=...) -> DROutput."""
    install_shims()
    from openmatch.modeling.dense_retrieval_model import DRModelForInference
    lm = build_reference_model(cfg, state_dict)
    m = DRModelForInference.__new__(DRModelForInference)
    import torch
    torch.nn.Module.__init__(m)
    m.lm_q = lm
    m.lm_p = lm
    m.head_q = None
    m.head_p = None
    m.tied = True
    m.feature = "last_hidden_state"
    m.pooling = pooling
    m.normalize = True
    m.model_args = None
    m.train_args = None
    m.data_args = None
    return m.eval()
