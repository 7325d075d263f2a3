// Dense bf16 GEMM with fused epilogues for every linear layer of the VisRAG-Ret encode path
// (reference call sites: SURVEY.md section 2b, K2,K5,K7,K8,K9,K11,K12,K15,K17,K18):
//   vision_transformer.py:79-105 (qkv, proj), mlp.py:34-47 (fc1/GELU/fc2),
//   patch_embed.py:65,87 (conv == GEMM over im2col rows), resampler.py:124,128,166-167,
//   modeling_minicpm.py:850-871 (q/k/v + RoPE :259-290), :908,983-985 (o_proj + scaled
//   residual), :293-335 (SwiGLU MLP).
// Main loop: gemm_core.h.  Roofline: MFMA (bf16 dense 2.5 PFLOP/s).
#include "gemm_core.h"
#include "kernels.h"

namespace vr {

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));   // nn.GELU() default (exact)
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

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
    const int fr = lane & 15, fq = lane >> 4;

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        if (m >= p.M) continue;
        const int orow = p.rowmap ? p.rowmap[m] : m;
        if (orow < 0) continue;
        const float* rb = nullptr;
        if (p.rowbias) rb = p.rowbias + (size_t)(m % p.rowbias_period) * p.rowbias_ld;

        if constexpr (EPI == EPI_SWIGLU) {
            // W rows are interleaved in blocks of 16: [16 gate | 16 up | ...]; fragment j even
            // is gate, j odd is up, for the same 16 output columns.
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int oc = (n0 + wn * 64) / 2 + jj * 16 + fq * 4;
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    o[r] = f2bf(silu(acc[i][2 * jj][r]) * acc[i][2 * jj + 1][r]);
                *reinterpret_cast<bf16x4*>((bf16_t*)p.out + (size_t)orow * p.ldo + oc) = o;
            }
        } else if constexpr (EPI == EPI_ROPE) {
            // the wave's 64 columns are exactly one head (head_dim 64): rotate-half pairs
            // (c, c+32) live in fragments (j, j+2) of the same lane.  fp32, like
            // apply_rotary_pos_emb (modeling_minicpm.py:259-290); table = [pos][32] cos | [32] sin.
            const int nb = n0 + wn * 64;
            if (nb < p.rope_cols) {
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
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = nb + j * 16 + fq * 4;
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(acc[i][j][r]);
                *reinterpret_cast<bf16x4*>((bf16_t*)p.out + (size_t)orow * p.ldo + n) = o;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + fq * 4;
                f32x4 v = acc[i][j];
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                if (rb && n < p.rowbias_cols) v += *reinterpret_cast<const f32x4*>(rb + n);
                if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU) {
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        o[r] = f2bf(EPI == EPI_GELU ? gelu_erf(v[r]) : v[r]);
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
}

template <int EPI>
static hipError_t launch_epi(const GemmArgs& a, int variant, hipStream_t s) {
    const int tiles = (a.N / GEMM_BN) * ((a.M + GEMM_BM - 1) / GEMM_BM);
    if (variant == 0x100) {   // experiment: same kernel at 1 workgroup per CU (130 KiB of LDS requested)
        auto k = gemm_bf16_kernel<EPI, true>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024);
        hipLaunchKernelGGL(k, dim3(tiles), dim3(256), 130 * 1024, s, a);
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_SMEM_BYTES);
        return hipGetLastError();
    }
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
