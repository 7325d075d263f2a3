"""-m gpu: every HIP kernel, called through the C ABI, against an fp32 restatement of the same
op on the same (bf16-rounded) inputs.  Tolerances are stated per test."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.gpu_util import op_attention, op_gemm, op_norm  # noqa: E402

DEV = "cuda:0"


def _bf(t):
    return t.to(torch.bfloat16)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale)


# ------------------------------------------------------------------------------- GEMM ---
@pytest.mark.parametrize("variant", [0, 3, 9, 12])
@pytest.mark.parametrize("M,N,K", [(128, 256, 64), (300, 256, 192), (1024, 512, 640), (77, 256, 2304), (700, 768, 128),
                                   (4500, 4096, 512)])   # 288 tiles of 256 x 256: the persistent, dynamically scheduled form
def test_gemm_bias_bf16(M, N, K, variant):
    A, W, b = _bf(_rand((M, K), 1)), _bf(_rand((N, K), 2, 0.1)), _rand((N,), 3)
    ref = A.float() @ W.float().T + b
    out = op_gemm(A.to(DEV), W.to(DEV), 0, bias=b.to(DEV), variant=variant).float().cpu()
    # fp32 accumulate, one bf16 rounding of the output: |err| <= 2^-8 |ref| + accumulation noise
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-2, atol=2e-2)
    # transpose-detecting: asymmetric inputs, elementwise agreement much tighter than a swap
    assert (out - ref).abs().max() < 0.05 * ref.abs().max()


@pytest.mark.parametrize("M,N,K", [(300, 192, 64), (1000, 1152, 320), (520, 384, 4352), (11000, 1152, 512)])
def test_gemm_192_tile(M, N, K):
    """256x192 tile kernel (variant 7): bf16 / GELU / fp32 / residual epilogues."""
    A, W, b = _bf(_rand((M, K), 41)), _bf(_rand((N, K), 42, 0.1)), _rand((N,), 43)
    acc = A.float() @ W.float().T + b
    Ad, Wd, bd = A.to(DEV), W.to(DEV), b.to(DEV)
    np.testing.assert_allclose(op_gemm(Ad, Wd, 0, bias=bd, variant=7).float().cpu().numpy(), acc.numpy(), rtol=1e-2, atol=2e-2)
    np.testing.assert_allclose(op_gemm(Ad, Wd, 1, bias=bd, variant=7).float().cpu().numpy(),
                               torch.nn.functional.gelu(acc).numpy(), rtol=1e-2, atol=2e-2)
    r = _rand((M, N), 44)
    for variant in (7, 13):             # 13: the one-wave-per-SIMD form of the tile (residual and fp32 epilogues)
        out = op_gemm(Ad, Wd, 3, bias=bd, resid=r.to(DEV), alpha=0.5, out_dtype=torch.float32, variant=variant).cpu()
        np.testing.assert_allclose(out.numpy(), (r + 0.5 * acc).numpy(), rtol=1e-5, atol=1e-4)
        out = op_gemm(Ad, Wd, 2, bias=bd, out_dtype=torch.float32, variant=variant).cpu()
        np.testing.assert_allclose(out.numpy(), acc.numpy(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("M,N,K", [(100, 192, 64), (128, 384, 128), (2176, 2304, 2304), (2176, 2304, 5760), (1000, 1152, 192), (129, 768, 320)])
def test_gemm_128w_tile(M, N, K):
    """Half-height one-wave tiles (gemm128w.hip, variants 14 = 128 x 192, 15 = 128 x 256): residual in place on both, the
    lookup-free bf16 / GELU / SwiGLU forms on the 256-wide one; K of 1, 2, 3, 5 and 36 / 90 steps walks every exit of the
    three-stage loop; the decoder's o / down shapes."""
    A, W, b = _bf(_rand((M, K), 61)), _bf(_rand((N, K), 62, 0.1)), _rand((N,), 63)
    acc = A.float() @ W.float().T + b
    Ad, Wd, bd = A.to(DEV), W.to(DEV), b.to(DEV)
    r = _rand((M, N), 64)
    ref = (r + 0.5 * acc).numpy()
    tol = dict(rtol=1e-5, atol=1e-4 * max(1.0, K / 512))
    out = op_gemm(Ad, Wd, 3, bias=bd, resid=r.to(DEV), alpha=0.5, out_dtype=torch.float32, variant=14).cpu()
    np.testing.assert_allclose(out.numpy(), ref, **tol)
    # ... bit-identical to the 256-row kernel's result: the same MFMA chain per output, whatever the tile
    out13 = op_gemm(Ad, Wd, 3, bias=bd, resid=r.to(DEV), alpha=0.5, out_dtype=torch.float32, variant=13).cpu()
    assert torch.equal(out, out13)
    if N % 256 == 0:
        out = op_gemm(Ad, Wd, 3, bias=bd, resid=r.to(DEV), alpha=0.5, out_dtype=torch.float32, variant=15).cpu()
        np.testing.assert_allclose(out.numpy(), ref, **tol)
        assert torch.equal(op_gemm(Ad, Wd, 0, bias=bd, variant=15), op_gemm(Ad, Wd, 0, bias=bd, variant=12))
        assert torch.equal(op_gemm(Ad, Wd, 1, bias=bd, variant=15), op_gemm(Ad, Wd, 1, bias=bd, variant=12))
        np.testing.assert_allclose(op_gemm(Ad, Wd, 1, bias=bd, variant=15).float().cpu().numpy(), torch.nn.functional.gelu(acc).numpy(), rtol=1e-2, atol=2e-2 * max(1.0, K / 512))
        I = N // 2
        il = lambda g, u: torch.stack([g.reshape(I // 16, 16, *g.shape[1:]), u.reshape(I // 16, 16, *u.shape[1:])], dim=1).reshape(2 * I, *g.shape[1:])  # noqa: E731
        Wi, bi = il(W[:I], W[I:]).to(DEV), il(b[:I], b[I:]).to(DEV)
        sw = torch.nn.functional.silu(acc[:, :I]) * acc[:, I:]
        o15 = op_gemm(Ad, Wi, 4, bias=bi, out_cols=I, variant=15)
        np.testing.assert_allclose(o15.float().cpu().numpy(), sw.numpy(), rtol=1e-2, atol=3e-2 * max(1.0, K / 512))
        assert torch.equal(o15, op_gemm(Ad, Wi, 4, bias=bi, out_cols=I, variant=12))
    else:
        with pytest.raises(Exception):
            op_gemm(Ad, Wd, 3, bias=bd, resid=r.to(DEV), alpha=0.5, out_dtype=torch.float32, variant=15)


def test_gemm_identity_asymmetric():
    """A = I (padded), asymmetric W: catches row/col swaps of the MFMA C layout."""
    K = 128
    A = torch.eye(K)[:100]
    W = (torch.arange(256 * K).reshape(256, K) % 97).float() / 97.0 + torch.arange(256)[:, None] * 0.01
    out = op_gemm(_bf(A).to(DEV), _bf(W).to(DEV), 2, out_dtype=torch.float32).cpu()
    np.testing.assert_allclose(out.numpy(), _bf(W).float().T[:100].numpy(), rtol=0, atol=1e-6)


@pytest.mark.parametrize("variant", [0, 3, 9, 12])
@pytest.mark.parametrize("M,N,K", [(200, 256, 128), (300, 384, 64)])     # (384: the 256-wide tiles' column edge — the general forms)
def test_gemm_epilogues(variant, M, N, K):
    A, W, b = _bf(_rand((M, K), 4)), _bf(_rand((N, K), 5, 0.2)), _rand((N,), 6)
    acc = A.float() @ W.float().T
    Ad, Wd, bd = A.to(DEV), W.to(DEV), b.to(DEV)
    # 1: exact-erf GELU
    out = op_gemm(Ad, Wd, 1, bias=bd, variant=variant).float().cpu()
    np.testing.assert_allclose(out.numpy(), torch.nn.functional.gelu(acc + b).numpy(), rtol=1e-2, atol=2e-2)
    # 2: fp32 out
    out = op_gemm(Ad, Wd, 2, bias=bd, out_dtype=torch.float32, variant=variant).cpu()
    np.testing.assert_allclose(out.numpy(), (acc + b).numpy(), rtol=1e-5, atol=1e-4)
    # 3: residual with scale
    r = _rand((M, N), 7)
    out = op_gemm(Ad, Wd, 3, bias=bd, resid=r.to(DEV), alpha=0.2214, out_dtype=torch.float32, variant=variant).cpu()
    np.testing.assert_allclose(out.numpy(), (r + 0.2214 * (acc + b)).numpy(), rtol=1e-5, atol=1e-4)
    # 4: SwiGLU with 16-row interleaved [gate|up] weights
    I = N // 2
    Wg, Wu = W[:I], W[I:]
    Wi = torch.stack([Wg.reshape(I // 16, 16, K), Wu.reshape(I // 16, 16, K)], dim=1).reshape(N, K)
    ref = torch.nn.functional.silu(A.float() @ Wg.float().T) * (A.float() @ Wu.float().T)
    out = op_gemm(Ad, Wi.to(DEV), 4, out_cols=I, variant=variant).float().cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-2, atol=3e-2)


@pytest.mark.parametrize("variant", [0, 3, 9, 12])
@pytest.mark.parametrize("M,I,K", [(200, 128, 128), (1500, 3456, 1280)])
def test_gemm_swiglu_with_bias(variant, M, I, K):
    """The vision tower's MLP: gate / up projections WITH bias, bias interleaved like the weight rows."""
    A = _bf(_rand((M, K), 51))
    Wg, Wu = _bf(_rand((I, K), 52, 0.05)), _bf(_rand((I, K), 53, 0.05))
    bg, bu = _rand((I,), 54, 0.5), _rand((I,), 55, 0.5)
    il = lambda g, u: torch.stack([g.reshape(I // 16, 16, *g.shape[1:]), u.reshape(I // 16, 16, *u.shape[1:])], dim=1).reshape(2 * I, *g.shape[1:])  # noqa: E731
    ref = torch.nn.functional.silu(A.float() @ Wg.float().T + bg) * (A.float() @ Wu.float().T + bu)
    out = op_gemm(A.to(DEV), il(Wg, Wu).to(DEV), 4, bias=il(bg, bu).to(DEV), out_cols=I, variant=variant).float().cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-2, atol=3e-2)


@pytest.mark.parametrize("variant", [0, 3, 9, 12])
def test_gemm_rope(variant):
    """EPI_ROPE == apply_rotary_pos_emb (modeling_minicpm.py:259-290) on q,k columns; v untouched."""
    M, E, K = 150, 256, 128            # 4 heads of 64; N = 3E
    A, W = _bf(_rand((M, K), 8)), _bf(_rand((3 * E, K), 9, 0.2))
    pos = torch.arange(M, dtype=torch.int32) % 50
    inv = 1.0 / (10000.0 ** (torch.arange(0, 64, 2).float() / 64))
    fr = torch.outer(torch.arange(64).float(), inv)
    table = torch.cat([fr.cos(), fr.sin()], dim=1).contiguous()          # [pos][cos32|sin32]
    acc = (A.float() @ W.float().T)
    ref = acc.clone()
    for part in range(2):                                                # q and k
        for h in range(E // 64):
            x = acc[:, part * E + h * 64: part * E + (h + 1) * 64]
            c, s = fr.cos()[pos.long()], fr.sin()[pos.long()]
            x1, x2 = x[:, :32], x[:, 32:]
            ref[:, part * E + h * 64: part * E + h * 64 + 32] = x1 * c - x2 * s
            ref[:, part * E + h * 64 + 32: part * E + (h + 1) * 64] = x2 * c + x1 * s
    out = op_gemm(A.to(DEV), W.to(DEV), 5, rope_pos=pos.to(DEV), rope_table=table.to(DEV), rope_cols=2 * E,
                  variant=variant).float().cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-2, atol=3e-2)


# ------------------------------------------------------------------------------ norms ---
@pytest.mark.parametrize("dim,ldo", [(1152, 1152), (2304, 2304), (288, 384)])
def test_layernorm(dim, ldo):
    x, w, b = _rand((37, dim), 10, 3.0) + 0.5, _rand((dim,), 11) * 0.2 + 1, _rand((dim,), 12) * 0.1
    ref = torch.nn.functional.layer_norm(x, (dim,), w, b, 1e-6)
    out = op_norm(0, x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, ldo).float().cpu()
    np.testing.assert_allclose(out[:, :dim].numpy(), ref.numpy(), rtol=8e-3, atol=8e-3)   # bf16 output
    assert (out[:, dim:] == 0).all()


def test_rmsnorm():
    dim = 2304
    x, w = _rand((45, dim), 13, 2.0), _rand((dim,), 14) * 0.2 + 1
    ref = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * w        # modeling_minicpm.py:119-123
    out = op_norm(1, x.to(DEV), w.to(DEV), None, 1e-5).float().cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=8e-3, atol=8e-3)


# -------------------------------------------------------------------------- attention ---
def _ref_attn(q, k, v, causal, scale):
    s = (q.float() @ k.float().T) * scale
    if causal:
        L, Lk = s.shape
        s = s + torch.full((L, Lk), float("-inf")).triu(1)
    return torch.softmax(s, -1) @ v.float()


@pytest.mark.parametrize("hd,heads,lens,causal", [
    (72, 3, [64, 64], False), (72, 2, [1024], False), (72, 2, [1026, 60], False),
    (64, 3, [68, 13, 130, 1], True), (64, 2, [700], True), (128, 2, [1024, 60], False),
    (80, 2, [1024, 60], False), (80, 3, [64, 32, 64, 16, 4, 240], False), (80, 2, [700, 130], True)])   # 80: the EVisRAG vision tower
def test_attention(hd, heads, lens, causal):
    q_shared = hd == 128
    B = len(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    T = int(cu[-1])
    W = heads * hd
    qkv = _bf(_rand((T, 3 * W), 20, 1.0))
    scale = hd ** -0.5
    if q_shared:
        Qn = 64
        qs = _bf(_rand((Qn, W), 21))
        cu_q = torch.arange(B + 1, dtype=torch.int32) * Qn
        out = op_attention(qs.to(DEV), qkv[:, W:2 * W].to(DEV), qkv[:, 2 * W:].to(DEV), cu_q.to(DEV), cu.to(DEV), heads,
                           hd, Qn, False, True, scale, B * Qn).float().cpu()
    else:
        d = qkv.to(DEV)
        out = op_attention(d[:, :W], d[:, W:2 * W], d[:, 2 * W:], cu.to(DEV), cu.to(DEV), heads, hd, max(lens), causal,
                           False, scale, T).float().cpu()
    for b in range(B):
        lo, hi = int(cu[b]), int(cu[b + 1])
        for h in range(heads):
            k = qkv[lo:hi, W + h * hd: W + (h + 1) * hd]
            v = qkv[lo:hi, 2 * W + h * hd: 2 * W + (h + 1) * hd]
            if q_shared:
                ref = _ref_attn(qs[:, h * hd:(h + 1) * hd], k, v, False, scale)
                got = out[b * 64:(b + 1) * 64, h * hd:(h + 1) * hd]
            else:
                ref = _ref_attn(qkv[lo:hi, h * hd:(h + 1) * hd], k, v, causal, scale)
                got = out[lo:hi, h * hd:(h + 1) * hd]
            # P and the output are rounded to bf16: 2^-8 relative on O(1) values
            np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("lens", [[1024], [1024, 1000, 777, 256], [512, 300, 33], [256], [1280, 1], [960, 1023, 65]])
def test_attention_vit_one_wave_per_simd(lens):
    """attention_w.hip (head_dim 72, non-causal, query tiles of 256 rows, the hand-ordered stream): full tiles, ragged key
    counts (masked last tile, junk halves), sequences shorter than one query tile next to long ones."""
    hd, heads = 72, 3
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    T, W = int(cu[-1]), heads * hd
    qkv = _bf(_rand((T, 3 * W), 70 + len(lens), 1.0))
    d = qkv.to(DEV)
    scale = hd ** -0.5
    out = op_attention(d[:, :W], d[:, W:2 * W], d[:, 2 * W:], cu.to(DEV), cu.to(DEV), heads, hd, max(lens), False, False,
                       scale, T).float().cpu()
    for b in range(len(lens)):
        lo, hi = int(cu[b]), int(cu[b + 1])
        for h in range(heads):
            ref = _ref_attn(qkv[lo:hi, h * hd:(h + 1) * hd], qkv[lo:hi, W + h * hd:W + (h + 1) * hd],
                            qkv[lo:hi, 2 * W + h * hd:2 * W + (h + 1) * hd], False, scale)
            np.testing.assert_allclose(out[lo:hi, h * hd:(h + 1) * hd].numpy(), ref.numpy(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("case", ["late_spikes", "rising", "all_negative", "huge_first"])
def test_attention_vit_running_maximum_is_raised_correctly(case):
    """The rare block of attention_w.hip that raises a row's running maximum between two periods: O, the packed P not yet
    multiplied and the scores already computed against the old maximum must each be rescaled exactly once
    (cdna guide T13).  Inputs that force it in early, middle and late tiles, for single rows and for all rows at once."""
    hd, L = 72, 1024
    q = _rand((L, hd), 40); k = _rand((L, hd), 41); v = _rand((L, hd), 42)
    if case == "late_spikes":            # single keys dominate single rows, in tiles 3, 9 and 15 and inside both halves of a tile
        for row, key, gain in ((10, 250, 8.0), (300, 600, 12.0), (301, 633, 6.0), (777, 1000, 10.0), (1023, 1023, 9.0), (0, 40, 7.0)):
            k[key] = q[row] * gain
    elif case == "rising":               # every row's maximum grows tile after tile
        k = k * torch.linspace(0.2, 6.0, L)[:, None]
        k = q.mean(0, keepdim=True) * torch.linspace(0.0, 3.0, L)[:, None] + k
    elif case == "all_negative":         # scores far below zero everywhere (the first half sets a negative maximum)
        k = -q.mean(0, keepdim=True).expand(L, hd) * 4.0 + 0.1 * k
        q = q + 2.0 * q.mean(0, keepdim=True)
    else:                                # the first key towers over everything after it
        k[0] = q.mean(0) * 30.0
    qkv = _bf(torch.cat([q, k, v], dim=1))
    cu = torch.tensor([0, L], dtype=torch.int32)
    d = qkv.to(DEV)
    out = op_attention(d[:, :hd], d[:, hd:2 * hd], d[:, 2 * hd:], cu.to(DEV), cu.to(DEV), 1, hd, L, False, False,
                       hd ** -0.5, L).float().cpu()
    # Called on its own the kernel multiplies q by scale * log2(e) and rounds the product to bf16 (inside the model that
    # factor rides the qkv GEMM's epilogue, GemmArgs::col_scale: one rounding).  With keys twelve times the usual size the
    # second rounding moves a logit by up to 0.05 — the reference models it, so that the test stays about the rescaling.
    qs = _bf(qkv[:, :hd].float() * (hd ** -0.5 * 1.4426950408889634)).float()
    ref = torch.softmax(qs @ qkv[:, hd:2 * hd].float().T * 0.6931471805599453, -1) @ qkv[:, 2 * hd:].float()
    assert torch.isfinite(out).all()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("lens,causal", [([68] * 7, True), ([68, 13, 80, 1, 16, 17, 79, 33], True), ([68, 5, 80, 48], False)])
def test_attention_short_sequences_one_wave_each(lens, causal):
    """attention_small.hip: self-attention of packed sequences of at most 80 tokens (head_dim 64, cu_q IS cu_kv — the
    decoder over a page's 68 tokens), one wave per (sequence, head), against the fp32 reference and against the tiled
    kernel (which the same call takes when the two offset arrays are different buffers)."""
    hd, heads = 64, 5
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    T, W = int(cu[-1]), heads * hd
    qkv = _bf(_rand((T, 3 * W), 30, 1.0))
    d, cud = qkv.to(DEV), cu.to(DEV)
    scale = hd ** -0.5
    small = op_attention(d[:, :W], d[:, W:2 * W], d[:, 2 * W:], cud, cud, heads, hd, max(lens), causal, False, scale, T).float().cpu()
    tiled = op_attention(d[:, :W], d[:, W:2 * W], d[:, 2 * W:], cud, cud.clone(), heads, hd, max(lens), causal, False, scale, T).float().cpu()
    for b in range(len(lens)):
        lo, hi = int(cu[b]), int(cu[b + 1])
        for h in range(heads):
            ref = _ref_attn(qkv[lo:hi, h * hd:(h + 1) * hd], qkv[lo:hi, W + h * hd: W + (h + 1) * hd],
                            qkv[lo:hi, 2 * W + h * hd: 2 * W + (h + 1) * hd], causal, scale)
            np.testing.assert_allclose(small[lo:hi, h * hd:(h + 1) * hd].numpy(), ref.numpy(), rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(small.numpy(), tiled.numpy(), rtol=2e-2, atol=1e-2)


def test_attention_segments_head_dim_128():
    """The vision tower's form: head_dim-128 slots, bidirectional attention inside ragged row segments (windows of up
    to 64 rows, whole images), own q rows per segment, a scale that is not 1/sqrt(128)."""
    hd, heads = 128, 3
    for lens in ([64, 32, 64, 16, 4, 240], [700, 64]):
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
        T, W = int(cu[-1]), heads * hd
        qkv = _bf(_rand((T, 3 * W), 60, 1.0))
        scale = 80 ** -0.5
        d = qkv.to(DEV)
        out = op_attention(d[:, :W], d[:, W:2 * W], d[:, 2 * W:], cu.to(DEV), cu.to(DEV), heads, hd, max(lens), False, False,
                           scale, T).float().cpu()
        for b in range(len(lens)):
            lo, hi = int(cu[b]), int(cu[b + 1])
            for h in range(heads):
                ref = _ref_attn(qkv[lo:hi, h * hd:(h + 1) * hd], qkv[lo:hi, W + h * hd:W + (h + 1) * hd],
                                qkv[lo:hi, 2 * W + h * hd:2 * W + (h + 1) * hd], False, scale)
                np.testing.assert_allclose(out[lo:hi, h * hd:(h + 1) * hd].numpy(), ref.numpy(), rtol=2e-2, atol=2e-2)


def test_attention_spiked_scores():
    """Force the online-softmax rescale branch: one key dominates from a late tile."""
    hd, heads, L = 64, 1, 300
    q = _rand((L, hd), 30); k = _rand((L, hd), 31); v = _rand((L, hd), 32)
    k[250] = q[10] * 8.0                                # huge logit for row 10 in the 4th tile
    qkv = _bf(torch.cat([q, k, v], dim=1))
    cu = torch.tensor([0, L], dtype=torch.int32)
    d = qkv.to(DEV)
    out = op_attention(d[:, :hd], d[:, hd:2 * hd], d[:, 2 * hd:], cu.to(DEV), cu.to(DEV), 1, hd, L, False, False,
                       hd ** -0.5, L).float().cpu()
    ref = _ref_attn(qkv[:, :hd], qkv[:, hd:2 * hd], qkv[:, 2 * hd:], False, hd ** -0.5)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=2e-2, atol=2e-2)


def test_stream_pairs_for_two_batches_in_flight_overlap():
    """engine.overlapping_streams: the pair of streams DRModelForInference keeps two batches in flight on must not share a
    hardware queue (vr_streams_overlap probes with two spin kernels); a stream never overlaps itself."""
    from visrag_amd.engine import overlapping_streams, streams_overlap
    sa, sb = overlapping_streams(0, 2)
    assert sa.cuda_stream != sb.cuda_stream
    assert streams_overlap(sa, sb)
    assert not streams_overlap(sa, sa)
    three = overlapping_streams(0, 3)
    assert len({s.cuda_stream for s in three}) == 3
