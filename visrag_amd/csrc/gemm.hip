// Dense bf16 GEMM with fused epilogues for every linear layer of the VisRAG-Ret encode path
// (reference call sites: SURVEY.md section 2b, K2,K5,K7,K8,K9,K11,K12,K15,K17,K18):
//   vision_transformer.py:79-105 (qkv, proj), mlp.py:34-47 (fc1/GELU/fc2),
//   patch_embed.py:65,87 (conv == GEMM over im2col rows), resampler.py:124,128,166-167,
//   modeling_minicpm.py:850-871 (q/k/v + RoPE :259-290), :908,983-985 (o_proj + scaled
//   residual), :293-335 (SwiGLU MLP).
// Main loops: gemm_core.h (128x128 4-wave tile; 256x256 8-wave tile).  Roofline: MFMA.
#include "gemm_core.h"
#include "gemm_core_mid.h"
#include "gemm_core_stag.h"
#include "kernels.h"
#include <cstdlib>

namespace vr {

// nn.GELU() default = exact erf form (timm mlp.py / vision_transformer.py:466), NOT the tanh
// approximation.  erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the bf16
// rounding of the output): 1 rcp + 1 exp + 6 fma instead of libm erff's ~30 instructions, which
// cost the fc1 GEMM a quarter of its throughput.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
    const float erf_abs = fmaf(-poly * t, e, 1.0f);          // erf(|x|/sqrt2)
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// Epilogue of one 16-row fragment strip: this lane holds out[m][nb + j*16 + fq*4 + r], j = 0..3.
// `nb` is the first of the wave's 64 output columns.
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_row(f32x4 (&acc)[4], const GemmArgs& p, int m, int nb, int fq) {
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
        for (int j = 0; j < 4; ++j) {
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

template <int EPI, bool GLDS>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = p.N / GEMM_BN;
    const int tiles_m = (p.M + GEMM_BM - 1) / GEMM_BM;
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (t / tiles_n) * GEMM_BM, n0 = (t % tiles_n) * GEMM_BN;

    gemm_acc_t acc;
    gemm_zero(acc);
    gemm_mainloop<GLDS>(acc, (const bf16_t*)p.A, p.lda, (const bf16_t*)p.W, p.ldw, m0, n0, p.K, smem);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        gemm_epilogue_row<EPI>(acc[i], p, m0 + wm * 64 + i * 16 + (lane & 15), n0 + wn * 64, lane >> 4);
}

// 256x256 tile, 8 waves (see gemm_core.h).  W must have readable rows up to the next multiple of
// 256 (the engine pads weights); columns >= N are not stored.
template <int EPI, int MODE>   // MODE 0: 2-stage, 1: BK32 x 4 stages, 2: 2-stage with mid-tile prefetch
__global__ __launch_bounds__(512, 2) void gemm256_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = (p.N + G256_BN - 1) / G256_BN;
    const int tiles_m = (p.M + G256_BM - 1) / G256_BM;
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    // Grouped rasterisation: the ~32 tiles an XCD runs concurrently (consecutive t) form a patch of
    // GM m-tiles x 4 n-tiles, so every A / W k-slice fetched into that XCD's L2 is reused by 4 / GM
    // workgroups (row-major order would make it 1 A + 32 W slices per step: ~half the requests miss).
    // Measured (8192^3): GM=1 1135 TF, GM=4 1243 TF; the K=1152 ViT shapes move < 3 %.
    const int GM = p.raster_gm > 0 ? p.raster_gm : 1;
    const int gsz = GM * tiles_n;
    const int g = t / gsz, r = t % gsz;
    const int gm = min(GM, tiles_m - g * GM);         // last group may be shorter
    const int m0 = (g * GM + r % gm) * G256_BM, n0 = (r / gm) * G256_BN;

    gemm256_acc_t acc;
    gemm256_zero(acc);
    if constexpr (MODE == 1) gemm256_mainloop_p4(acc, (const bf16_t*)p.A, p.lda, (const bf16_t*)p.W, p.ldw, m0, n0, p.K, smem);
    else if constexpr (MODE == 3) gemm256_mainloop_stag(acc, (const bf16_t*)p.A, p.lda, (const bf16_t*)p.W, p.ldw, m0, n0, p.K, smem);
    else if constexpr (MODE == 2) gemm256_mainloop_mid(acc, (const bf16_t*)p.A, p.lda, (const bf16_t*)p.W, p.ldw, m0, n0, p.K, smem);
    else gemm256_mainloop(acc, (const bf16_t*)p.A, p.lda, (const bf16_t*)p.W, p.ldw, m0, n0, p.K, smem);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        gemm_epilogue_row<EPI>(acc[i], p, m0 + wm * 128 + i * 16 + (lane & 15), n0 + wn * 64, lane >> 4);
}

template <int EPI>
static hipError_t launch_epi(const GemmArgs& a_in, int variant, hipStream_t s) {
    const GemmArgs& a = a_in;
    if (variant == GEMM_VARIANT_256 || variant == GEMM_VARIANT_256P4 || variant == GEMM_VARIANT_256MID ||
        variant == GEMM_VARIANT_256STAG) {
        GemmArgs a = a_in;   // (shadows the outer reference: raster_gm is filled in here)
        const int tn = (a.N + G256_BN - 1) / G256_BN;
        const int tiles = tn * ((a.M + G256_BM - 1) / G256_BM);
        static const int env_gm = getenv("VR_RASTER_GM") ? atoi(getenv("VR_RASTER_GM")) : 0;   // tuning aid
        if (env_gm > 0) a.raster_gm = env_gm;
        if (a.raster_gm <= 0) a.raster_gm = 4;   // sweep on MI355X: 4 is within noise of the best for every shape
        const int vi = (variant == GEMM_VARIANT_256) ? 0 : (variant == GEMM_VARIANT_256P4) ? 1
                     : (variant == GEMM_VARIANT_256MID) ? 2 : 3;
        void (*k)(GemmArgs) = vi == 0 ? gemm256_bf16_kernel<EPI, 0> : vi == 1 ? gemm256_bf16_kernel<EPI, 1>
                            : vi == 2 ? gemm256_bf16_kernel<EPI, 2> : gemm256_bf16_kernel<EPI, 3>;
        static bool attr[4] = {false, false, false, false};
        if (!attr[vi]) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G256_SMEM_BYTES); attr[vi] = true; }
        hipLaunchKernelGGL(k, dim3(tiles), dim3(512), G256_SMEM_BYTES, s, a);
        return hipGetLastError();
    }
    const int tiles = (a.N / GEMM_BN) * ((a.M + GEMM_BM - 1) / GEMM_BM);
    if (variant == GEMM_VARIANT_REG) {
        auto k = gemm_bf16_kernel<EPI, false>;
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_SMEM_BYTES); attr = true; }
        hipLaunchKernelGGL(k, dim3(tiles), dim3(256), GEMM_SMEM_BYTES, s, a);
    } else {
        auto k = gemm_bf16_kernel<EPI, true>;
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_SMEM_BYTES); attr = true; }
        hipLaunchKernelGGL(k, dim3(tiles), dim3(256), GEMM_SMEM_BYTES, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_gemm(const GemmArgs& a, int epi, int variant, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    if (variant == GEMM_VARIANT_AUTO) {
        // 256x256 tiles when M is big enough to fill the chip with them and N is a multiple of 256
        // or wide enough that one partial tile column costs little (3456 -> 14 tiles, +3.7 %);
        // N = 1152 (4.5 tiles) stays on the 128x128 kernel (measured: 806 vs 771 TF).
        // Small grids (decoder, M ~ 2k) are decided by wave quantisation: 256 one-per-CU slots for
        // the big tile (~1.25x faster per flop) vs 512 two-per-CU slots for the small one.
        const bool n_ok = (a.N % 256 == 0) || a.N >= 2048;
        const long t256 = (long)((a.N + 255) / 256) * ((a.M + 255) / 256);
        const long t128 = (long)(a.N / 128) * ((a.M + 127) / 128);
        const double e256 = 1.25 * (double)t256 / (double)(((t256 + 255) / 256) * 256);
        const double e128 = (double)t128 / (double)(((t128 + 511) / 512) * 512);
        variant = (n_ok && e256 > e128) ? GEMM_VARIANT_256 : GEMM_VARIANT_GLDS;
    }
    switch (epi) {
        case EPI_BF16: return launch_epi<EPI_BF16>(a, variant, s);
        case EPI_GELU: return launch_epi<EPI_GELU>(a, variant, s);
        case EPI_F32: return launch_epi<EPI_F32>(a, variant, s);
        case EPI_RESID: return launch_epi<EPI_RESID>(a, variant, s);
        case EPI_SWIGLU: return launch_epi<EPI_SWIGLU>(a, variant, s);
        case EPI_ROPE: return launch_epi<EPI_ROPE>(a, variant, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vr
