// PERSISTENT 256 x 256 x 64 GEMM: one workgroup per CU walks a list of output tiles.
//
// Ablations of the one-tile-per-workgroup kernel on the ViT shapes (K = 1152, 18 K-steps): with
// the main loop reduced to its MFMAs the launch still takes 0.211 ms against 0.145 ms of matrix
// work — ~9 us per 30 us tile go to workgroup dispatch, the first LDS-DMA round trip and the
// epilogue, during which the CU's matrix pipes idle (128 KiB of LDS = one workgroup per CU, so no
// other workgroup covers the gap).  Here the workgroup stays resident: the LDS-DMA of the NEXT
// tile's first K-step is issued before the epilogue of the current tile, so dispatch and the cold
// round trip are paid once per CU instead of once per tile and the epilogue's global stores
// overlap the next tile's loads.
// Tile order: XCD x (= blockIdx % 8) owns a contiguous range of the grouped-raster tile sequence
// and its 32 resident workgroups take consecutive tiles of it round after round, i.e. the same
// L2-friendly patches as the non-persistent kernel.
// Main loop: fragment reads interleaved with the MFMAs (gemm_core_il.h).
#include "gemm_core.h"
#include "gemm_core_il.h"
#include "gemm_epilogue.h"
#include "lab.h"
#include "gemm_core_lab.h"

namespace vr {

struct TileWalk {
    int tiles_m, tiles_n, GM;
    __device__ __forceinline__ void coords(int t, int& m0, int& n0) const {
        const int gsz = GM * tiles_n;
        const int g = t / gsz, r = t % gsz;
        const int gm = min(GM, tiles_m - g * GM);
        m0 = (g * GM + r % gm) * 256;
        n0 = (r / gm) * 256;
    }
};

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256p_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int arow = wm * 128 + fr, wrow = wn * 64 + fr;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* W = (const bf16_t*)p.W;

    TileWalk tw;
    tw.tiles_n = (p.N + 255) / 256;
    tw.tiles_m = (p.M + 255) / 256;
    tw.GM = p.raster_gm > 0 ? p.raster_gm : 1;
    const int ntiles = tw.tiles_m * tw.tiles_n;
    // this XCD's contiguous share of the tile sequence, and this workgroup's stride through it
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int q = ntiles >> 3, rem = ntiles & 7;
    const int base = (xcd < rem) ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    const int cnt = q + (xcd < rem ? 1 : 0);
    if (slot >= cnt) return;

    const int nk = p.K / GEMM_BK;
    constexpr int SB = 2 * G256_TILE_BYTES;
    int sp = 0;                                   // stage that holds (or is receiving) the next K-step
    int m0, n0;
    tw.coords(base + slot, m0, n0);
    stage_glds(A, p.lda, m0, 0, smem, wave, lane);
    stage_glds(W, p.ldw, n0, 0, smem + G256_TILE_BYTES, wave, lane);

    for (int ti = slot; ti < cnt; ti += per_xcd) {
        int nm0 = 0, nn0 = 0;
        const bool has_next = ti + per_xcd < cnt;
        if (has_next) tw.coords(base + ti + per_xcd, nm0, nn0);

        gemm256_acc_t acc;
        gemm256_zero(acc);
        for (int kt = 0; kt < nk; ++kt) {
            char* cur = smem + sp * SB;
            char* nxt = smem + (sp ^ 1) * SB;
            __syncthreads();                      // K-step landed; the other stage is free
            if (kt + 1 < nk) {
                stage_glds(A, p.lda, m0, (kt + 1) * GEMM_BK, nxt, wave, lane);
                stage_glds(W, p.ldw, n0, (kt + 1) * GEMM_BK, nxt + G256_TILE_BYTES, wave, lane);
            } else if (has_next) {                // first K-step of the NEXT tile flies under the epilogue
                stage_glds(A, p.lda, nm0, 0, nxt, wave, lane);
                stage_glds(W, p.ldw, nn0, 0, nxt + G256_TILE_BYTES, wave, lane);
            }
            const char* tA = cur;
            const char* tW = cur + G256_TILE_BYTES;
            bf16x8 w[4], a0, a1, wx[4], b0, b1;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = g256_frag(tW, wrow + j * 16, 0, fq);
            a0 = g256_frag(tA, arow, 0, fq);
            a1 = g256_frag(tA, arow + 16, 0, fq);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (g < 3) {
                        b0 = g256_frag(tA, arow + (2 * g + 2) * 16, kk, fq);
                        b1 = g256_frag(tA, arow + (2 * g + 3) * 16, kk, fq);
                    } else if (kk == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) wx[j] = g256_frag(tW, wrow + j * 16, 1, fq);
                        b0 = g256_frag(tA, arow, 1, fq);
                        b1 = g256_frag(tA, arow + 16, 1, fq);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[2 * g][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], a0, acc[2 * g][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[2 * g + 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], a1, acc[2 * g + 1][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    a0 = b0; a1 = b1;
                    if (g == 3 && kk == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) w[j] = wx[j];
                    }
                }
            }
            sp ^= 1;
        }
        // ---- epilogue (registers -> global); the next tile's first K-step is already in flight
        if constexpr (EPI == EPI_RESID) {
            if (!p.rowmap) gemm_epilogue_resid_tile<8, 4, 4>(acc, p, m0 + wm * 128 + fr, n0 + wn * 64, fq);
            else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    gemm_epilogue_row<EPI>(acc[i], p, m0 + wm * 128 + i * 16 + fr, n0 + wn * 64, fq);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                gemm_epilogue_row<EPI>(acc[i], p, m0 + wm * 128 + i * 16 + fr, n0 + wn * 64, fq);
        }
        m0 = nm0; n0 = nn0;
    }
}

template <int EPI>
static hipError_t launch256p_t(GemmArgs a, int n_cu, hipStream_t s) {
    const int tm = (a.M + 255) / 256, tn = (a.N + 255) / 256;
    if (a.raster_gm <= 0) a.raster_gm = tm <= 16 ? tm : 4;
    const int tiles = tm * tn;
    int grid = n_cu / 8 * 8;                       // one resident workgroup per CU, equal share per XCD
    if (grid > (tiles + 7) / 8 * 8) grid = (tiles + 7) / 8 * 8;
    auto k = gemm256p_bf16_kernel<EPI>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G256_SMEM_BYTES); attr = true; }
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), G256_SMEM_BYTES, s, a);
    return hipGetLastError();
}

hipError_t launch_gemm256p(const GemmArgs& a, int epi, hipStream_t s) {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        n_cu = prop.multiProcessorCount;
    }
    switch (epi) {
        case EPI_BF16: return launch256p_t<EPI_BF16>(a, n_cu, s);
        case EPI_GELU: return launch256p_t<EPI_GELU>(a, n_cu, s);
        case EPI_F32: return launch256p_t<EPI_F32>(a, n_cu, s);
        case EPI_RESID: return launch256p_t<EPI_RESID>(a, n_cu, s);
        case EPI_SWIGLU: return launch256p_t<EPI_SWIGLU>(a, n_cu, s);
        case EPI_ROPE: return launch256p_t<EPI_ROPE>(a, n_cu, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vr
