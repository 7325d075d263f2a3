"""Model dimensions of VisRAG-Ret (= MiniCPM-V-2.0: SigLIP-so400m ViT -> 64-query
resampler -> MiniCPM-2B decoder -> weighted-mean pool -> L2 norm).

Dimension provenance (reference file:line):
  * vision tower `vit_so400m_patch14_siglip_384` embed_dim=1152, depth=27 (last block
    dropped -> 26), heads=16, mlp_ratio=3.7362, no class token
    (timm_modified/timm/models/vision_transformer.py:2613-2619;
     src/openmatch/modeling/modeling_minicpmv/modeling_minicpmv.py:57-73)
  * resampler: query_num=64, heads = hidden//128
    (modeling_minicpmv.py:75-82; configuration_minicpm.py:201-224)
  * decoder sizes come from the HF checkpoint's config.json (not in the repo); the
    values below are the public MiniCPM-V-2.0 ones.
"""
from dataclasses import dataclass, asdict
import math


@dataclass
class VisRAGRetConfig:
    # vision tower
    patch_size: int = 14
    vit_dim: int = 1152
    vit_depth: int = 26            # 27 in the checkpoint, last block dropped
    vit_heads: int = 16            # head_dim = 72
    vit_mlp_ratio: float = 3.7362  # hidden = int(dim * ratio) = 4304
    vit_pos_grid: int = 27         # 384/14 -> 27x27 learned pos-embed, resampled per grid
    vit_ln_eps: float = 1e-6
    # resampler
    query_num: int = 64            # 8x8 queries
    resampler_ln_eps: float = 1e-6
    # decoder (MiniCPM-2B)
    hidden_size: int = 2304
    num_layers: int = 40
    num_heads: int = 36            # head_dim = 64
    intermediate_size: int = 5760
    vocab_size: int = 122753
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    scale_emb: float = 12.0
    scale_depth: float = 1.4
    # slicing policy (modeling_minicpmv.py:482-537)
    scale_resolution: int = 448
    max_slice_nums: int = 9
    slice_mode: bool = True
    # engine option (not a checkpoint field): token-only batches run the decoder at fp32-class precision
    # (hi + lo bf16 operand splits; csrc/hp_text.hip) — what keeps ~20-token queries inside the 1e-3 score bar
    # True / 1: every token-only batch; False / 0: never; N > 1: only batches whose longest sequence has <= N tokens
    # (e.g. 512, the reference's query length — long text passages then keep the bf16 MFMA attention path)
    text_split_precision: int = True
    # pooling of DRModel.encode (dense_retrieval_model.py:150-223): wmean | mean | lasttoken | cls | drop_wmean | drop_mean
    pooling: str = "wmean"

    @property
    def vit_head_dim(self) -> int:
        return self.vit_dim // self.vit_heads

    @property
    def vit_hidden(self) -> int:
        return int(self.vit_dim * self.vit_mlp_ratio)

    @property
    def resampler_heads(self) -> int:
        return self.hidden_size // 128

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def residual_scale(self) -> float:
        # modeling_minicpm.py:983-985
        return self.scale_depth / math.sqrt(self.num_layers)

    def to_dict(self):
        return asdict(self)

    # ---- analytic work model -------------------------------
    def flops_vit(self, n_patches: int) -> float:
        N, D, F = n_patches, self.vit_dim, self.vit_hidden
        pe = 2.0 * N * (3 * self.patch_size ** 2) * D
        blk = 2.0 * N * D * 3 * D + 4.0 * N * N * D + 2.0 * N * D * D + 4.0 * N * D * F
        return pe + self.vit_depth * blk

    def flops_resampler(self, n_patches: int) -> float:
        N, D, E, Q = n_patches, self.vit_dim, self.hidden_size, self.query_num
        return 2.0 * N * D * E + 4.0 * N * E * E + 6.0 * Q * E * E + 4.0 * Q * N * E

    def flops_decoder(self, L: int) -> float:
        E, I = self.hidden_size, self.intermediate_size
        return self.num_layers * (L * (8.0 * E * E + 6.0 * E * I) + 4.0 * L * L * E)

    def flops_page(self, n_patches: int = 1024, L: int = 68) -> float:
        return self.flops_vit(n_patches) + self.flops_resampler(n_patches) + self.flops_decoder(L)


def full_config() -> VisRAGRetConfig:
    return VisRAGRetConfig()


def tiny_config() -> VisRAGRetConfig:
    """Small dims with the SAME head sizes (72 / 128 / 64) so every kernel template is the
    production one; used by parity tests and the committed golden fixtures."""
    return VisRAGRetConfig(
        vit_dim=288, vit_depth=2, vit_heads=4, vit_pos_grid=6,
        hidden_size=256, num_layers=2, num_heads=4, intermediate_size=640,
        vocab_size=1000, scale_resolution=112,
    )
