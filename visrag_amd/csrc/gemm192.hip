// 256 (M) x 192 (N) x 64 tile for the N = 1152 GEMMs of the SigLIP tower (attn out-projection,
// MLP fc2: 1152 = 6 x 192, where 256-wide tiles would waste 10 % and the 128x128 kernel is bound by
// its LDS traffic).  512 threads = 8 waves as 4 (M) x 2 (N); a wave owns 64 x 96 = 4 x 6 fragments
// (96 fp32 accumulators); 10 fragment reads feed 24 MFMAs per K-half.  LDS: 2 stages x
// (A 32 KiB + W 24 KiB) = 112 KiB, one workgroup per CU.  Same LDS image / swizzle / LDS-DMA
// staging as gemm_core.h; grouped rasterisation as in gemm.hip.
#include "gemm_core.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace vr {

constexpr int G192_BM = 256, G192_BN = 192;
constexpr int G192_A_BYTES = 256 * 128, G192_W_BYTES = 192 * 128;
constexpr int G192_STAGE = G192_A_BYTES + G192_W_BYTES;      // 56 KiB
constexpr int G192_SMEM = 2 * G192_STAGE;                     // 112 KiB

// wave w fills rows [24w, 24w+24) of the W tile: 3 LDS-DMA instructions of 8 rows
__device__ __forceinline__ void stage_glds_w192(const bf16_t* __restrict__ g, int ld, int row0, int k0, char* tile,
                                                int wave, int lane) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int rbase = wave * 24 + t * 8;
        const int row = rbase + (lane >> 3);
        const int kc = (lane & 7) ^ (row & 7);
        const bf16_t* src = g + (size_t)(row0 + row) * ld + k0 + kc * 8;
        __builtin_amdgcn_global_load_lds(VR_GLOBAL(src), VR_LDS(tile + rbase * 128), 16, 0, 0);
    }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm192_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = p.N / G192_BN;
    const int tiles_m = (p.M + G192_BM - 1) / G192_BM;
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int GM = p.raster_gm > 0 ? p.raster_gm : 1;
    const int gsz = GM * tiles_n;
    const int g = t / gsz, r = t % gsz;
    const int gm = min(GM, tiles_m - g * GM);
    const int m0 = (g * GM + r % gm) * G192_BM, n0 = (r / gm) * G192_BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* W = (const bf16_t*)p.W;

    f32x4 acc[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / GEMM_BK;
    stage_glds(A, p.lda, m0, 0, smem, wave, lane);
    stage_glds_w192(W, p.ldw, n0, 0, smem + G192_A_BYTES, wave, lane);
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * G192_STAGE;
        char* nxt = smem + ((kt + 1) & 1) * G192_STAGE;
        __syncthreads();
        if (kt + 1 < nk) {
            stage_glds(A, p.lda, m0, (kt + 1) * GEMM_BK, nxt, wave, lane);
            stage_glds_w192(W, p.ldw, n0, (kt + 1) * GEMM_BK, nxt + G192_A_BYTES, wave, lane);
        }
        const char* tA = cur;
        const char* tW = cur + G192_A_BYTES;
        // fragment reads interleaved with the MFMAs (see gemm_core_il.h): strip i+1's A fragment —
        // or, in the last strip, the seven fragments that open the next K-half — is requested before
        // strip i's six MFMAs are issued
        auto frag = [&](const char* t, int row, int kk) {
            return *reinterpret_cast<const bf16x8*>(t + row * 128 + (((kk * 4 + fq) ^ (row & 7)) << 4));
        };
        const int arow = wm * 64 + fr, wrow = wn * 96 + fr;
        // (two A registers and one W set per K-half, all indices compile-time: no register copies)
        bf16x8 w[2][6], a[2];
#pragma unroll
        for (int j = 0; j < 6; ++j) w[0][j] = frag(tW, wrow + j * 16, 0);
        a[0] = frag(tA, arow, 0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = (kk * 4 + i) & 1, n = c ^ 1;
                if (i < 3) {
                    a[n] = frag(tA, arow + (i + 1) * 16, kk);
                } else if (kk == 0) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) w[1][j] = frag(tW, wrow + j * 16, 1);
                    a[n] = frag(tA, arow, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[kk][j], a[c], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if constexpr (EPI == EPI_RESID) {
        gemm_epilogue_resid_tile<4, 6, 4>(acc, p, m0 + wm * 64 + fr, n0 + wn * 96, fq);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            gemm_epilogue_row<EPI, 6>(acc[i], p, m0 + wm * 64 + i * 16 + fr, n0 + wn * 96, fq);
    }
}

template <int EPI>
static hipError_t launch192_t(GemmArgs a, hipStream_t s) {
    if (a.raster_gm <= 0) a.raster_gm = 4;
    const int tiles = (a.N / G192_BN) * ((a.M + G192_BM - 1) / G192_BM);
    auto k = gemm192_bf16_kernel<EPI>;
    static unsigned long long attr = 0;     // bit d: set on device d
    set_max_dynamic_lds((const void*)k, G192_SMEM, attr);
    hipLaunchKernelGGL(k, dim3(tiles), dim3(512), G192_SMEM, s, a);
    return hipGetLastError();
}

hipError_t launch_gemm192(const GemmArgs& a, int epi, hipStream_t s) {
    if (a.N % G192_BN || a.K % GEMM_BK) return hipErrorInvalidValue;
    switch (epi) {
        case EPI_BF16: return launch192_t<EPI_BF16>(a, s);
        case EPI_GELU: return launch192_t<EPI_GELU>(a, s);
        case EPI_F32: return launch192_t<EPI_F32>(a, s);
        case EPI_RESID: return launch192_t<EPI_RESID>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vr
