// Fused GEMM epilogues shared by every tile configuration (see gemm.hip for the call sites).
#pragma once
#include "common.h"
#include "kernels.h"

namespace vr {

// nn.GELU() default = exact erf form (timm mlp.py / vision_transformer.py:466), NOT the tanh
// approximation:  gelu(x) = 0.5 * x * (1 + erf(x / sqrt2)).
// The epilogue VALU work is not hidden by anything (one workgroup per CU: when its waves leave the K-loop the
// matrix pipe idles).  PMC on the fc1 GEMM (K = 1152, 18 K-steps): a VALU instruction occupies its SIMD's
// issue for 4 cycles, the 128 outputs per lane x 2 waves per SIMD of this epilogue came to ~20 % of the
// tile's time with a compare / select / |x| formulation (11.5 VALU + 3 s_nop per value).  So: erf as an ODD
// polynomial of the CLAMPED argument — no |x|, no compare, no select — in operations the compiler emits as
// packed pairs (v_pk_fma_f32 / v_pk_mul_f32: two values per issue slot):
//     xc = clamp(x, -X0, X0), X0 = 3 sqrt2;   erf(x / sqrt2) ~= xc * P(xc^2)   (1/sqrt2 folded into P)
//     gelu = hx + hx * e,  hx = x / 2
// 7 VALU slots per value.  |gelu error| <= 5e-5 for |x| <= 4.5 (degree-8 fit of erf, 2.2e-5) and
// <= 1.1e-5 |x| beyond (the clamped erf stops at 0.99998 instead of 1) — below the bf16 rounding of the stored
// activation wherever the output is not ~0, and ~1e-4 absolute on the zero side at |x| ~ 10.
__device__ __forceinline__ float gelu_erf(float x) {
    constexpr float X0 = 4.2426406871192851f;
    const float xc = fminf(fmaxf(x, -X0), X0);
    const float u = xc * xc;
    float p = 1.125353832e-10f;
    p = fmaf(p, u, -1.074373991e-08f);
    p = fmaf(p, u, 4.536592962e-07f);
    p = fmaf(p, u, -1.129243078e-05f);
    p = fmaf(p, u, 1.871812565e-04f);
    p = fmaf(p, u, -2.218800626e-03f);
    p = fmaf(p, u, 1.963623831e-02f);
    p = fmaf(p, u, -1.326938473e-01f);
    p = fmaf(p, u, 7.978062550e-01f);
    const float e = xc * p;                       // erf(x / sqrt2), saturating at +-0.99998
    const float hx = 0.5f * x;
    return fmaf(hx, e, hx);
}
// the same on four values at once: every Horner step is two independent packed FMAs, so consecutive steps
// issue back to back (a lone dependent v_pk_fma_f32 chain needs a wait state between its links)
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 x) {
    constexpr float X0 = 4.2426406871192851f;
    const f32x4 lo = {-X0, -X0, -X0, -X0}, hi = {X0, X0, X0, X0};
    const f32x4 xc = __builtin_elementwise_min(__builtin_elementwise_max(x, lo), hi);
    const f32x4 u = xc * xc;
    auto c4 = [](float c) { return f32x4{c, c, c, c}; };
    f32x4 p = c4(1.125353832e-10f);
    p = __builtin_elementwise_fma(p, u, c4(-1.074373991e-08f));
    p = __builtin_elementwise_fma(p, u, c4(4.536592962e-07f));
    p = __builtin_elementwise_fma(p, u, c4(-1.129243078e-05f));
    p = __builtin_elementwise_fma(p, u, c4(1.871812565e-04f));
    p = __builtin_elementwise_fma(p, u, c4(-2.218800626e-03f));
    p = __builtin_elementwise_fma(p, u, c4(1.963623831e-02f));
    p = __builtin_elementwise_fma(p, u, c4(-1.326938473e-01f));
    p = __builtin_elementwise_fma(p, u, c4(7.978062550e-01f));
    const f32x4 e = xc * p;
    const f32x4 hx = x * c4(0.5f);
    return __builtin_elementwise_fma(hx, e, hx);
}
// silu(x) = x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division: `x / (1 + e^-x)` costs a dozen
// instructions per value (v_div_scale x 2, v_rcp, four FMAs, v_div_fmas, v_div_fixup) — 1 500 of the 2 500 per lane and tile
// of the SwiGLU epilogue; the result is rounded to bf16 right after.  (x = -inf would give NaN where the division gave -0;
// finite x of any size is fine: e^-x overflows to inf, the reciprocal is 0.)
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

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
            f32x4 g = acc[2 * jj], u = acc[2 * jj + 1];
            if (p.bias) {                        // interleaved like the W rows: [16 gate | 16 up | ...]
                g += *reinterpret_cast<const f32x4*>(p.bias + nb + jj * 32 + fq * 4);
                u += *reinterpret_cast<const f32x4*>(p.bias + nb + jj * 32 + 16 + fq * 4);
            }
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(silu(g[r]) * u[r]);
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
                if constexpr (EPI == EPI_GELU) v = gelu_erf4(v);
                if constexpr (EPI == EPI_BF16) { if (n < p.col_scale_n) v *= p.col_scale; }        // (GemmArgs::col_scale: per column)
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
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

namespace vr {

// bf16-output epilogues of a whole wave tile (MI x 16 rows, 64 accumulator columns) with
// COALESCED stores.  In the MFMA C layout a lane owns 4 consecutive columns of one row, so a
// direct store instruction writes 16 rows x 32 B — sixteen partial-line requests per instruction;
// on the K = 1152 GEMMs the stores of the 256 x 256 tile cost ~11 us of a ~40 us tile (ablation:
// the same kernel without its stores runs at 1.84 instead of 1.18 PFLOP/s).  Here the wave first
// parks its converted tile in its private slice of the (now idle) LDS stages, swizzled like the
// operand tiles, and reads it back row-wise: one store instruction = 8 rows x 128 B, full lines.
//   wl: this wave's LDS slice, MI * 2 KiB (row pitch 128 B = 64 bf16; SwiGLU rows hold 32).
template <int EPI, int MI>
__device__ __forceinline__ void gemm_epilogue_tile_lds_general(f32x4 (&acc)[MI][4], const GemmArgs& p, int mrow0, int nb,
                                                               int lane, char* wl) {
    static_assert(EPI == EPI_BF16 || EPI == EPI_GELU || EPI == EPI_SWIGLU || EPI == EPI_ROPE, "bf16 outputs only");
    const int fr = lane & 15, fq = lane >> 4;
    // ---- 1. fused math in registers, bf16 tile into LDS
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = i * 16 + fr;
        const int m = min(mrow0 + row, p.M - 1);               // (rows >= M are never stored)
        char* lrow = wl + row * 128;
        if constexpr (EPI == EPI_SWIGLU) {
            // fragment j even = gate, j odd = up for the same 16 output columns (see gemm_epilogue_row)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                f32x4 g = acc[i][2 * jj], u = acc[i][2 * jj + 1];
                if (p.bias && nb + jj * 32 < p.N) {
                    g += *reinterpret_cast<const f32x4*>(p.bias + nb + jj * 32 + fq * 4);
                    u += *reinterpret_cast<const f32x4*>(p.bias + nb + jj * 32 + 16 + fq * 4);
                }
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(silu(g[r]) * u[r]);
                const int c = jj * 2 + (fq >> 1);              // 16-byte chunk of the 64-byte output row
                *reinterpret_cast<bf16x4*>(lrow + ((c ^ (row & 3)) << 4) + (fq & 1) * 8) = o;
            }
        } else {
            if constexpr (EPI == EPI_ROPE) {
                if (nb < p.rope_cols && nb < p.N) {
                    const float* tab = p.rope_table + (size_t)p.rope_pos[m] * 64;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 cs = *reinterpret_cast<const f32x4*>(tab + j * 16 + fq * 4);
                        const f32x4 sn = *reinterpret_cast<const f32x4*>(tab + 32 + j * 16 + fq * 4);
                        const f32x4 x1 = acc[i][j], x2 = acc[i][j + 2];
                        acc[i][j] = x1 * cs - x2 * sn;
                        acc[i][j + 2] = x2 * cs + x1 * sn;
                    }
                }
            }
            const float* rb = nullptr;
            if constexpr (EPI != EPI_ROPE) {
                if (p.rowbias) rb = p.rowbias + (size_t)(m % p.rowbias_period) * p.rowbias_ld;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = min(nb + j * 16 + fq * 4, p.N - 4);
                f32x4 v = acc[i][j];
                if constexpr (EPI != EPI_ROPE) {
                    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                    if (rb && n < p.rowbias_cols) v += *reinterpret_cast<const f32x4*>(rb + n);
                }
                if constexpr (EPI == EPI_GELU) v = gelu_erf4(v);
                if constexpr (EPI == EPI_BF16) { if (n < p.col_scale_n) v *= p.col_scale; }        // (GemmArgs::col_scale: per column)
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
                const int c = j * 2 + (fq >> 1);               // 16-byte chunk of the 128-byte row
                *reinterpret_cast<bf16x4*>(lrow + ((c ^ (row & 7)) << 4) + (fq & 1) * 8) = o;
            }
        }
    }
    // ---- 2. row-wise read-back (same wave: LDS operations of a wave execute in order), full-line stores
    bf16_t* out = (bf16_t*)p.out;
    if constexpr (EPI == EPI_SWIGLU) {
        // 64-byte output rows: 4 lanes per row, 16 rows per instruction
#pragma unroll
        for (int it = 0; it < MI; ++it) {
            const int row = it * 16 + (lane >> 2), c = lane & 3;
            const u32x4 d = *reinterpret_cast<const u32x4*>(wl + row * 128 + ((c ^ (row & 3)) << 4));
            const int m = mrow0 + row, n = nb / 2 + c * 8;
            if (m < p.M && 2 * n < p.N) {
                const int orow = p.rowmap ? p.rowmap[m] : m;
                if (orow >= 0) *reinterpret_cast<u32x4*>(out + (size_t)orow * p.ldo + n) = d;
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < MI * 2; ++it) {
            const int row = it * 8 + (lane >> 3), c = lane & 7;
            const u32x4 d = *reinterpret_cast<const u32x4*>(wl + row * 128 + ((c ^ (row & 7)) << 4));
            const int m = mrow0 + row, n = nb + c * 8;
            if (m < p.M && n < p.N) {
                const int orow = p.rowmap ? p.rowmap[m] : m;
                if (orow >= 0) *reinterpret_cast<u32x4*>(out + (size_t)orow * p.ldo + n) = d;
            }
        }
    }
}

// The same for the GEMMs that have no per-row lookups (no row map, no row bias: every ViT / decoder GEMM of
// the encode path).  In the general version each (strip, fragment) pair sits behind its own branches and
// loads its bias again (the staging stores may alias it as far as hipcc can tell), so a wave walks a chain of
// 32 dependent global-load round trips: in-kernel timestamps put the bf16 epilogue of a 256 x 256 tile at
// 12.4 us even with four workgroups on an idle chip.  Here everything a lane needs from memory is requested
// before the first use (bias: 4 loads per wave tile; RoPE: all positions, then the cos/sin rows two strips at
// a time), the staging buffer is addressed as LDS, and the store loop has no loads at all.
template <int EPI, int MI>
__device__ __forceinline__ void gemm_epilogue_tile_lds_plain(f32x4 (&acc)[MI][4], const GemmArgs& p, int mrow0, int nb,
                                                             int lane, char* wl_generic) {
    typedef __attribute__((address_space(3))) char* lds_p;
    const lds_p wl = (lds_p)VR_LDS(wl_generic);
    const int fr = lane & 15, fq = lane >> 4;
    if constexpr (EPI == EPI_SWIGLU) {
        f32x4 bias[4];                           // interleaved like the W rows: fragment j even = gate, odd = up
#pragma unroll
        for (int j = 0; j < 4; ++j) bias[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bias[j] = *reinterpret_cast<const f32x4*>(p.bias + min(nb + j * 16 + fq * 4, p.N - 4));
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = i * 16 + fr;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const f32x4 g = acc[i][2 * jj] + bias[2 * jj], u = acc[i][2 * jj + 1] + bias[2 * jj + 1];
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(silu(g[r]) * u[r]);
                const int c = jj * 2 + (fq >> 1);
                *reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(wl + row * 128 + ((c ^ (row & 3)) << 4) + (fq & 1) * 8) = o;
            }
        }
    } else {
        f32x4 bias[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bias[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU) {
            if (p.bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bias[j] = *reinterpret_cast<const f32x4*>(p.bias + min(nb + j * 16 + fq * 4, p.N - 4));
            }
        }
        if constexpr (EPI == EPI_ROPE) {
            if (nb < p.rope_cols && nb < p.N) {        // (wave-uniform) this 64-column block is a q or k head
                int pos[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) pos[i] = p.rope_pos[min(mrow0 + i * 16 + fr, p.M - 1)];
#pragma unroll
                for (int c = 0; c < MI; c += 2) {
                    f32x4 cs[2][2], sn[2][2];
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const float* tab = p.rope_table + (size_t)pos[c + ii] * 64;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            cs[ii][j] = *reinterpret_cast<const f32x4*>(tab + j * 16 + fq * 4);
                            sn[ii][j] = *reinterpret_cast<const f32x4*>(tab + 32 + j * 16 + fq * 4);
                        }
                    }
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const f32x4 x1 = acc[c + ii][j], x2 = acc[c + ii][j + 2];
                            acc[c + ii][j] = x1 * cs[ii][j] - x2 * sn[ii][j];
                            acc[c + ii][j + 2] = x2 * cs[ii][j] + x1 * sn[ii][j];
                        }
                }
            }
        }
        if constexpr (EPI == EPI_BF16) {
            if (nb < p.col_scale_n) {                  // (wave-uniform: this 64-column block is scaled — the ViT's q heads)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bias[j] *= p.col_scale;
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[i][j] *= p.col_scale;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = i * 16 + fr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = acc[i][j] + bias[j];
                if constexpr (EPI == EPI_GELU) v = gelu_erf4(v);
                const bf16x4 o = __builtin_convertvector(v, bf16x4);       // two v_cvt_pk_bf16_f32
                const int c = j * 2 + (fq >> 1);
                *reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(wl + row * 128 + ((c ^ (row & 7)) << 4) + (fq & 1) * 8) = o;
            }
        }
    }
    // row-wise read-back (same wave: its LDS operations execute in order), full-line stores, no lookups
    bf16_t* out = (bf16_t*)p.out;
    if constexpr (EPI == EPI_SWIGLU) {
        u32x4 d[MI];
#pragma unroll
        for (int it = 0; it < MI; ++it) {
            const int row = it * 16 + (lane >> 2), c = lane & 3;
            d[it] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(wl + row * 128 + ((c ^ (row & 3)) << 4));
        }
#pragma unroll
        for (int it = 0; it < MI; ++it) {
            const int row = it * 16 + (lane >> 2), c = lane & 3;
            const int m = mrow0 + row, n = nb / 2 + c * 8;
            if (m < p.M && 2 * n < p.N) *reinterpret_cast<u32x4*>(out + (size_t)m * p.ldo + n) = d[it];
        }
    } else {
        // (all read-backs first: behind the predicated stores hipcc issues them one at a time, an LDS round trip each)
        u32x4 d[MI * 2];
#pragma unroll
        for (int it = 0; it < MI * 2; ++it) {
            const int row = it * 8 + (lane >> 3), c = lane & 7;
            d[it] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(wl + row * 128 + ((c ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int it = 0; it < MI * 2; ++it) {
            const int row = it * 8 + (lane >> 3), c = lane & 7;
            const int m = mrow0 + row, n = nb + c * 8;
            if (m < p.M && n < p.N) *reinterpret_cast<u32x4*>(out + (size_t)m * p.ldo + n) = d[it];
        }
    }
}

// The lookup-free epilogue as the one-wave-per-SIMD kernel runs it (round 5): the bias of the wave's columns is loaded ONCE per
// tile by the caller (a buffer load whose descriptor is empty without a bias and ends at N: no branch, no clamp), and the
// rows leave through a BUFFER store — the descriptor covers the tile's valid rows (rows >= M are dropped by the range
// check), a column past N turns the lane's offset out of range.  No predicates, no exec-mask branches, no 64-bit address
// arithmetic: the general form above spends ~1 400 instructions per lane and tile, a third of them on exactly that.
//   ors: output rows [m0, min(m0 + 256, M)) of the tile;  row_off: byte offset of the piece's first row in it
//   rbrs / rb_off (bf16 epilogue only): the per-row bias table f32 [period][rowbias_ld] behind a descriptor and the byte offset of
//   the piece's first row in it — the launcher admits a row bias here only when period % 256 == 0 (a tile never wraps) and
//   rowbias_cols % 64 == 0; the caller passes an EMPTY descriptor otherwise (loads return 0)
template <int EPI, int MI>
__device__ __forceinline__ void gemm_epilogue_tile_lds_plain_buf(f32x4 (&acc)[MI][4], f32x4 (&bias)[4], const GemmArgs& p,
                                                                 __amdgpu_buffer_rsrc_t ors, unsigned row_off, unsigned ldo_bytes,
                                                                 const f32x4 (&rope)[MI][4], int nb, int lane, char* wl_generic,
                                                                 __amdgpu_buffer_rsrc_t rbrs, unsigned rb_off) {
    static_assert(EPI == EPI_BF16 || EPI == EPI_GELU || EPI == EPI_SWIGLU || EPI == EPI_ROPE, "bf16 outputs only");
    typedef __attribute__((address_space(3))) char* lds_p;
    const lds_p wl = (lds_p)VR_LDS(wl_generic);
    const int fr = lane & 15, fq = lane >> 4;
    if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = i * 16 + fr;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const f32x4 g = acc[i][2 * jj] + bias[2 * jj], u = acc[i][2 * jj + 1] + bias[2 * jj + 1];
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(silu(g[r]) * u[r]);
                const int c = jj * 2 + (fq >> 1);
                *reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(wl + row * 128 + ((c ^ (row & 3)) << 4) + (fq & 1) * 8) = o;
            }
        }
    } else {
        if constexpr (EPI == EPI_ROPE) {
            // rope[i] = cos[0:16 | 16:32] and sin[0:16 | 16:32] of row i's position at this lane's four columns — loaded by the caller a
            // strip group ahead (the same for every head): with the position and table loads inside this branch every piece waited
            // for a round trip of its own (and for the previous piece's stores: one counter) — 8 per tile, ~11 us of a 61 us tile
            if (nb < p.rope_cols && nb < p.N) {        // (wave-uniform) this 64-column block is a q or k head
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 x1 = acc[i][j], x2 = acc[i][j + 2];
                        acc[i][j] = x1 * rope[i][j] - x2 * rope[i][2 + j];
                        acc[i][j + 2] = x2 * rope[i][j] + x1 * rope[i][2 + j];
                    }
            }
        }
        if constexpr (EPI == EPI_BF16) {
            if (p.rowbias && nb < p.rowbias_cols) {    // (wave-uniform: the resampler's k half gets pos_k[row % patches])
                f32x4 rb[MI][4];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        rb[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            rbrs, rb_off + (unsigned)(i * 16 + fr) * (unsigned)p.rowbias_ld * 4u + (unsigned)(nb + j * 16 + fq * 4) * 4u, 0, 0));
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] += rb[i][j];
            }
            if (nb < p.col_scale_n) {                  // (wave-uniform: this 64-column block is scaled — the ViT's q heads)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[i][j] = (acc[i][j] + bias[j]) * p.col_scale;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < MI; ++i) acc[i][j] += bias[j];
            }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = i * 16 + fr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = acc[i][j];
                if constexpr (EPI == EPI_GELU) v = gelu_erf4(v + bias[j]);
                const bf16x4 o = __builtin_convertvector(v, bf16x4);       // two v_cvt_pk_bf16_f32
                const int c = j * 2 + (fq >> 1);
                *reinterpret_cast<__attribute__((address_space(3))) bf16x4*>(wl + row * 128 + ((c ^ (row & 7)) << 4) + (fq & 1) * 8) = o;
            }
        }
    }
    // row-wise read-back (same wave: its LDS operations execute in order), full-line buffer stores
    constexpr unsigned OOR = 0x80000000u;              // beyond every descriptor
    if constexpr (EPI == EPI_SWIGLU) {
        u32x4 d[MI];
#pragma unroll
        for (int it = 0; it < MI; ++it) {
            const int row = it * 16 + (lane >> 2), c = lane & 3;
            d[it] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(wl + row * 128 + ((c ^ (row & 3)) << 4));
        }
        const int n = nb / 2 + (lane & 3) * 8;
        const unsigned cofs = 2 * n < p.N ? (unsigned)n * 2u : OOR;
#pragma unroll
        for (int it = 0; it < MI; ++it)
            __builtin_amdgcn_raw_buffer_store_b128(d[it], ors, row_off + (unsigned)(it * 16 + (lane >> 2)) * ldo_bytes + cofs, 0, 0);
    } else {
        u32x4 d[MI * 2];
#pragma unroll
        for (int it = 0; it < MI * 2; ++it) {
            const int row = it * 8 + (lane >> 3), c = lane & 7;
            d[it] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(wl + row * 128 + ((c ^ (row & 7)) << 4));
        }
        const int n = nb + (lane & 7) * 8;
        const unsigned cofs = n < p.N ? (unsigned)n * 2u : OOR;
#pragma unroll
        for (int it = 0; it < MI * 2; ++it)
            __builtin_amdgcn_raw_buffer_store_b128(d[it], ors, row_off + (unsigned)(it * 8 + (lane >> 3)) * ldo_bytes + cofs, 0, 0);
    }
}

template <int EPI, int MI>
__device__ __forceinline__ void gemm_epilogue_tile_lds(f32x4 (&acc)[MI][4], const GemmArgs& p, int mrow0, int nb,
                                                       int lane, char* wl) {
    if (!p.rowmap && !p.rowbias) gemm_epilogue_tile_lds_plain<EPI, MI>(acc, p, mrow0, nb, lane, wl);
    else gemm_epilogue_tile_lds_general<EPI, MI>(acc, p, mrow0, nb, lane, wl);
}

}  // namespace vr
