// Fused GEMM epilogues shared by every tile configuration (see gemm.hip for the call sites).
#pragma once
#include "common.h"
#include "kernels.h"

namespace vr {

// nn.GELU() default = exact erf form (timm mlp.py / vision_transformer.py:466), NOT the tanh
// approximation:  gelu(x) = 0.5 * (x + |x| * erf(|x| / sqrt2)).
// The epilogue VALU work is not hidden by anything (one workgroup per CU: when its waves leave
// the K-loop the matrix pipe idles), and for the fc1 GEMM (K = 1152, 18 K-steps) an erf with one
// v_rcp + one v_exp per element (quarter-rate transcendentals: Abramowitz-Stegun 7.1.26, and far
// worse libm erff) cost ~9 us per 256 x 256 tile against ~18 us of MFMA work.  Here erf is a pure
// FMA chain that the compiler packs two elements per instruction (v_pk_fma_f32):
//     erf(z) ~= z * P(z^2),  |z| <= 3   (degree-8 minimax-style fit, |err| <= 2.2e-5);   1 beyond
// (1 - erf(3) = 2.2e-5).  |gelu error| <= 5.3e-5 absolute everywhere — an order of magnitude below
// the bf16 rounding of the stored activation (half ulp >= 2.4e-4 for |y| >= 0.0625).
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = fabsf(x);
    const float z = ax * 0.70710678118654752440f;
    const float zc = fminf(z, 3.0f);
    const float t = zc * zc;
    float p = 4.074216068e-08f;
    p = fmaf(p, t, -1.944824664e-06f);
    p = fmaf(p, t, 4.106055228e-05f);
    p = fmaf(p, t, -5.110370805e-04f);
    p = fmaf(p, t, 4.235428344e-03f);
    p = fmaf(p, t, -2.510286350e-02f);
    p = fmaf(p, t, 1.110793381e-01f);
    p = fmaf(p, t, -3.753148771e-01f);
    p = fmaf(p, t, 1.128268426e+00f);
    const float e = (z >= 3.0f) ? 1.0f : zc * p;              // erf(|x| / sqrt2)
    return 0.5f * fmaf(ax, e, x);
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// Epilogue of one 16-row fragment strip: this lane holds out[m][nb + j*16 + fq*4 + r], j = 0..3.
// `nb` is the first of the wave's 64 output columns.
template <int EPI, int NF = 4>
__device__ __forceinline__ void gemm_epilogue_row(f32x4 (&acc)[NF], const GemmArgs& p, int m, int nb, int fq) {
    static_assert(NF == 4 || (EPI != EPI_SWIGLU && EPI != EPI_ROPE), "SwiGLU / RoPE epilogues need 64-column wave tiles");
    if (m >= p.M) return;
    const int orow = p.rowmap ? p.rowmap[m] : m;
    if (orow < 0) return;
    if constexpr (EPI == EPI_SWIGLU) {
        // W rows are interleaved in blocks of 16: [16 gate | 16 up | ...]; fragment j even is
        // gate, j odd is up, for the same 16 output columns.
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if (nb + jj * 32 >= p.N) continue;
            const int oc = nb / 2 + jj * 16 + fq * 4;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(silu(acc[2 * jj][r]) * acc[2 * jj + 1][r]);
            *reinterpret_cast<bf16x4*>((bf16_t*)p.out + (size_t)orow * p.ldo + oc) = o;
        }
    } else if constexpr (EPI == EPI_ROPE) {
        // the wave's 64 columns are exactly one head (head_dim 64): rotate-half pairs (c, c+32)
        // live in fragments (j, j+2) of the same lane.  fp32, like apply_rotary_pos_emb
        // (modeling_minicpm.py:259-290); table = [pos][32 cos | 32 sin].
        if (nb >= p.N) return;
        if (nb < p.rope_cols) {
            const float* tab = p.rope_table + (size_t)p.rope_pos[m] * 64;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 cs = *reinterpret_cast<const f32x4*>(tab + j * 16 + fq * 4);
                const f32x4 sn = *reinterpret_cast<const f32x4*>(tab + 32 + j * 16 + fq * 4);
                const f32x4 x1 = acc[j], x2 = acc[j + 2];
                acc[j] = x1 * cs - x2 * sn;
                acc[j + 2] = x2 * cs + x1 * sn;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nb + j * 16 + fq * 4;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(acc[j][r]);
            *reinterpret_cast<bf16x4*>((bf16_t*)p.out + (size_t)orow * p.ldo + n) = o;
        }
    } else {
        const float* rb = nullptr;
        if (p.rowbias) rb = p.rowbias + (size_t)(m % p.rowbias_period) * p.rowbias_ld;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int n = nb + j * 16 + fq * 4;
            if (n >= p.N) continue;
            f32x4 v = acc[j];
            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
            if (rb && n < p.rowbias_cols) v += *reinterpret_cast<const f32x4*>(rb + n);
            if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(EPI == EPI_GELU ? gelu_erf(v[r]) : v[r]);
                *reinterpret_cast<bf16x4*>((bf16_t*)p.out + (size_t)orow * p.ldo + n) = o;
            } else if constexpr (EPI == EPI_F32) {
                *reinterpret_cast<f32x4*>((float*)p.out + (size_t)orow * p.ldo + n) = v;
            } else {   // EPI_RESID: out = resid + alpha * (acc + bias); may alias resid
                const f32x4 rs = *reinterpret_cast<const f32x4*>(p.resid + (size_t)orow * p.ldo + n);
                *reinterpret_cast<f32x4*>((float*)p.out + (size_t)orow * p.ldo + n) = rs + p.alpha * v;
            }
        }
    }
}

}  // namespace vr

namespace vr {

// Residual epilogue for a whole wave tile: out = resid + alpha * (acc + bias), fp32, in place.
// All residual fragments of a chunk are LOADED FIRST with unconditional (row/column-clamped)
// addresses, then combined and stored under a predicate.  The per-row version above puts each
// row's loads behind a divergent `m < M` branch, so hipcc drains vmcnt(0) at every join and the
// fp32 read-modify-write of the residual stream (8 B/element, the epilogue's whole cost) runs
// with one row of loads in flight; here MI_CH x NF 16-byte loads are in flight per lane.
template <int MI, int NF, int MI_CH>
__device__ __forceinline__ void gemm_epilogue_resid_tile(f32x4 (&acc)[MI][NF], const GemmArgs& p, int mrow0,
                                                         int nb, int fq) {
    static_assert(MI % MI_CH == 0, "chunking");
    const float* __restrict__ resid = p.resid;
    float* __restrict__ out = (float*)p.out;
    int ncol[NF];
    f32x4 bias[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int n = nb + j * 16 + fq * 4;
        ncol[j] = n;
        const int nc = min(n, p.N - 4);
        bias[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < MI; c += MI_CH) {
        f32x4 rs[MI_CH][NF];
#pragma unroll
        for (int ii = 0; ii < MI_CH; ++ii) {
            const size_t ro = (size_t)min(mrow0 + (c + ii) * 16, p.M - 1) * p.ldo;
#pragma unroll
            for (int j = 0; j < NF; ++j)
                rs[ii][j] = *reinterpret_cast<const f32x4*>(resid + ro + min(ncol[j], p.N - 4));
        }
#pragma unroll
        for (int ii = 0; ii < MI_CH; ++ii) {
            const int m = mrow0 + (c + ii) * 16;
            if (m < p.M) {
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    if (ncol[j] < p.N)
                        *reinterpret_cast<f32x4*>(out + (size_t)m * p.ldo + ncol[j]) =
                            rs[ii][j] + p.alpha * (acc[c + ii][j] + bias[j]);
            }
        }
    }
}

}  // namespace vr
