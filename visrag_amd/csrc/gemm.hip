// Dense bf16 GEMM with fused epilogues for every linear layer of the VisRAG-Ret encode path
// (reference call sites: SURVEY.md section 2b, K2,K5,K7,K8,K9,K11,K12,K15,K17,K18):
//   vision_transformer.py:79-105 (qkv, proj), mlp.py:34-47 (fc1/GELU/fc2),
//   patch_embed.py:65,87 (conv == GEMM over im2col rows), resampler.py:124,128,166-167,
//   modeling_minicpm.py:850-871 (q/k/v + RoPE :259-290), :908,983-985 (o_proj + scaled
//   residual), :293-335 (SwiGLU MLP).
// Main loops: gemm_core.h (128x128 4-wave tile; 256x256 8-wave tile).  Roofline: MFMA.
#include "gemm_core.h"
#include "gemm_core_il.h"
#include "kernels.h"
#include "gemm_epilogue.h"

namespace vr {

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = p.N / GEMM_BN;
    const int tiles_m = (p.M + GEMM_BM - 1) / GEMM_BM;
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    // grouped rasterisation (see gemm256_bf16_kernel): with few m-tiles (decoder, M ~ 2k) GM = all of
    // them, i.e. n-major order, so a W tile is fetched once and shared by every m-tile instead of W
    // being streamed once per m-row (PMC: 904 MB fetched for a 63 MB gate/up GEMM before this).
    const int GM = p.raster_gm > 0 ? p.raster_gm : 1;
    const int gsz = GM * tiles_n;
    const int g = t / gsz, r = t % gsz;
    const int gm = min(GM, tiles_m - g * GM);
    const int m0 = (g * GM + r % gm) * GEMM_BM, n0 = (r / gm) * GEMM_BN;

    gemm_acc_t acc;
    gemm_zero(acc);
    gemm_mainloop(acc, (const bf16_t*)p.A, p.lda, (const bf16_t*)p.W, p.ldw, m0, n0, p.K, smem);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    if constexpr (EPI == EPI_RESID) {
        if (!p.rowmap) { gemm_epilogue_resid_tile<4, 4, 4>(acc, p, m0 + wm * 64 + (lane & 15), n0 + wn * 64, lane >> 4); return; }
    }
    if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU || EPI == EPI_SWIGLU || EPI == EPI_ROPE) {
        // coalesced stores through the idle LDS stages (the main loop ends with a barrier): 8 KiB per wave
        if ((p.N & 7) == 0 && (p.ldo & 7) == 0) {
            gemm_epilogue_tile_lds<EPI, 4>(acc, p, m0 + wm * 64, n0 + wn * 64, lane, smem + wave * 8192);
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        gemm_epilogue_row<EPI>(acc[i], p, m0 + wm * 64 + i * 16 + (lane & 15), n0 + wn * 64, lane >> 4);
}

// 256x256 tile, 8 waves (see gemm_core.h, gemm_core_il.h).  W must have readable rows up to the next multiple of
// 256 (the engine pads weights); columns >= N are not stored.
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = (p.N + G256_BN - 1) / G256_BN;
    const int tiles_m = (p.M + G256_BM - 1) / G256_BM;
    // split-K (EPI_F32 only): ksplit workgroups per tile, split s takes K-range [s, s+1) * K/ksplit and
    // writes its partial product to out + s * split_stride; the consumer sums them in a fixed order
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int tt = xcd_remap(blockIdx.x, tiles_m * tiles_n * ks);
    const int split = tt / (tiles_m * tiles_n);
    const int t = tt - split * (tiles_m * tiles_n);
    // Grouped rasterisation: the ~32 tiles an XCD runs concurrently (consecutive t) form a patch of
    // GM m-tiles x 4 n-tiles, so every A / W k-slice fetched into that XCD's L2 is reused by 4 / GM
    // workgroups (row-major order would make it 1 A + 32 W slices per step: ~half the requests miss).
    // Measured (8192^3): GM=1 1135 TF, GM=4 1243 TF; the K=1152 ViT shapes move < 3 %.
    const int GM = p.raster_gm > 0 ? p.raster_gm : 1;
    const int gsz = GM * tiles_n;
    const int g = t / gsz, r = t % gsz;
    const int gm = min(GM, tiles_m - g * GM);         // last group may be shorter
    const int m0 = (g * GM + r % gm) * G256_BM, n0 = (r / gm) * G256_BN;
    if (p.m_dev && m0 >= p.m_dev[0] - p.m_sub) return;       // row count known on the device only (the search's band pass)

    gemm256_acc_t acc;
    gemm256_zero(acc);
    const int Ks = p.K / ks;
    const bf16_t* Ap = (const bf16_t*)p.A + (size_t)split * Ks;
    const bf16_t* Wp = (const bf16_t*)p.W + (size_t)split * Ks;
    gemm256_mainloop_il(acc, Ap, p.lda, Wp, p.ldw, m0, n0, Ks, smem);

    if constexpr (EPI == EPI_F32) {
        if (ks > 1) {
            GemmArgs ps = p;
            ps.out = (float*)p.out + (size_t)split * p.split_stride;
            if (split > 0) ps.bias = nullptr;
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                gemm_epilogue_row<EPI_F32>(acc[i], ps, m0 + (wave >> 2) * 128 + i * 16 + (lane & 15), n0 + (wave & 3) * 64, lane >> 4);
            return;
        }
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    if constexpr (EPI == EPI_RESID) {
        if (!p.rowmap) { gemm_epilogue_resid_tile<8, 4, 4>(acc, p, m0 + wm * 128 + (lane & 15), n0 + wn * 64, lane >> 4); return; }
    }
    if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU || EPI == EPI_SWIGLU || EPI == EPI_ROPE) {
        // (every main loop ends with a workgroup barrier: the stages are free, 16 KiB per wave)
        if ((p.N & 7) == 0 && (p.ldo & 7) == 0) {
            gemm_epilogue_tile_lds<EPI, 8>(acc, p, m0 + wm * 128, n0 + wn * 64, lane, smem + wave * 16384);
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
        gemm_epilogue_row<EPI>(acc[i], p, m0 + wm * 128 + i * 16 + (lane & 15), n0 + wn * 64, lane >> 4);
}

template <int EPI>
static hipError_t launch_epi(const GemmArgs& a_in, int variant, hipStream_t s) {
    GemmArgs a = a_in;
    if (variant == GEMM_VARIANT_256IL) {
        const int tn = (a.N + G256_BN - 1) / G256_BN, tm = (a.M + G256_BM - 1) / G256_BM;
        if (a.raster_gm <= 0) a.raster_gm = tm <= 16 ? tm : 4;   // 4 is within noise of the best for big M; few m-tiles -> n-major
        auto k = gemm256_bf16_kernel<EPI>;
        static unsigned long long attr = 0;     // bit d: set on device d
        set_max_dynamic_lds((const void*)k, G256_SMEM_BYTES, attr);
        int grid = tn * tm;
        if (a.ksplit > 1) {
            if (EPI != EPI_F32 || a.K % (a.ksplit * GEMM_BK) || a.rowmap || a.rowbias) return hipErrorInvalidValue;
            grid *= a.ksplit;
        }
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), G256_SMEM_BYTES, s, a);
        return hipGetLastError();
    }
    if (variant != GEMM_VARIANT_GLDS) return hipErrorInvalidValue;
    if (a.ksplit > 1) return hipErrorInvalidValue;          // split-K exists on the 256-tile kernel only
    const int tiles = (a.N / GEMM_BN) * ((a.M + GEMM_BM - 1) / GEMM_BM);
    if (a.raster_gm <= 0) {
        const int tm = (a.M + GEMM_BM - 1) / GEMM_BM;
        a.raster_gm = tm <= 32 ? tm : 4;
    }
    auto k = gemm_bf16_kernel<EPI>;
    static unsigned long long attr = 0;     // bit d: set on device d
    set_max_dynamic_lds((const void*)k, GEMM_SMEM_BYTES, attr);
    hipLaunchKernelGGL(k, dim3(tiles), dim3(256), GEMM_SMEM_BYTES, s, a);
    return hipGetLastError();
}

hipError_t launch_gemm(const GemmArgs& a, int epi, int variant, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    // column scaling exists in the bf16-output epilogues only, on whole 64-column blocks (the tile forms test the block start)
    if (a.col_scale_n != 0 && (epi != EPI_BF16 || a.col_scale_n < 0 || a.col_scale_n % 64 != 0)) return hipErrorInvalidValue;
    if (variant == GEMM_VARIANT_192) return launch_gemm192(a, epi, s);
    if (variant == GEMM_VARIANT_256W) return launch_gemm256w(a, epi, s);
    if (variant == GEMM_VARIANT_192W) return launch_gemm192w(a, epi, s);
    if (variant == GEMM_VARIANT_128W_192) return launch_gemm128w(a, epi, 192, s);
    if (variant == GEMM_VARIANT_128W_256) return launch_gemm128w(a, epi, 256, s);
    if (variant == GEMM_VARIANT_AUTO) {
        // N = 1152 (SigLIP proj / fc2): 6 x 192 columns, big M -> the 256x192 kernel
#ifndef VR_GEMM_AUTO_192W
#define VR_GEMM_AUTO_192W 1
#endif
        if (a.N % 192 == 0 && a.N % 256 != 0 && a.N <= 1536 && a.M >= 4096 && epi <= EPI_RESID) {
            if (VR_GEMM_AUTO_192W && epi == EPI_RESID && !a.rowmap && gemm256w_fits(a, 192)) return launch_gemm192w(a, epi, s);
            return launch_gemm192(a, epi, s);
        }
        // 256x256 tiles when N is a multiple of 256 or wide enough that one partial tile column costs
        // little (3456 -> 14 tiles, +3.7 %); N = 1152 (4.5 tiles) goes to the 256x192 kernel above.
        // Small grids (decoder, M ~ 2k) are decided by wave quantisation: 256 one-per-CU slots for the
        // big tile vs 512 two-per-CU slots for the 128x128 one, whose flops cost ~1.45x the time and
        // ~1.25x the energy of the one-wave-per-SIMD kernel's (tools/op_energy.py, decoder gate/up:
        // 405 big tiles in two rounds 107 us / 134 mJ, 1530 small tiles in three rounds 121 us / 168 mJ).
        const bool n_ok = (a.N % 256 == 0) || a.N >= 2048;
        const long t256 = (long)((a.N + 255) / 256) * ((a.M + 255) / 256);
        const long t128 = (long)(a.N / 128) * ((a.M + 127) / 128);
        const double e256 = 1.45 * (double)t256 / (double)(((t256 + 255) / 256) * 256);
        const double e128 = (double)t128 / (double)(((t128 + 511) / 512) * 512);
#ifndef VR_GEMM_AUTO_W
#define VR_GEMM_AUTO_W 0xFF
#endif
        // Which 256x256 kernel: the one-wave-per-SIMD kernel (gemm256w.hip) is faster and leaner in isolation
        // everywhere (ViT qkv 230 vs 255 us, 302 vs 326 mJ), but its denser matrix-core stream pulls the shader
        // clock down (1.9 vs 2.1 GHz sustained, well below the 1400 W cap) and the clock recovers slowly, so in
        // the model the kernels that FOLLOW it lose part of what it gains: decide in-model, on one box
        // (tools/ab_libs.sh; bit e of VR_GEMM_AUTO_W puts epilogue e on the new kernel).  Round 2: everything on
        // it 48.1 ms/step, all but the ViT qkv GEMM 48.7, none (8-wave kernel, 128x128 gate/up) 50.5.
        const bool w_ok = (((VR_GEMM_AUTO_W) >> epi) & 1) && gemm256w_fits(a, 256);     // (else: the 8-wave kernel, 64-bit addresses)
        variant = (n_ok && e256 > e128) ? (w_ok ? GEMM_VARIANT_256W : GEMM_VARIANT_256IL) : GEMM_VARIANT_GLDS;
        if (variant == GEMM_VARIANT_256W) return launch_gemm256w(a, epi, s);
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
