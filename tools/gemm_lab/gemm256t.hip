// 256 x 256 x 64 GEMM with L2 TOUCH-AHEAD (variant 11).
//
// What bounds the shipped loop (gemm_core_il.h) is neither the MFMA pipe (MFMA-only: 1.9 PF) nor
// the L2 -> LDS path (LDS-DMA of L2-resident lines runs at 34 TB/s, tools/probe_l2bw.hip) but the
// 10-20 % of stage lines that MISS the 4 MiB per-XCD L2: with two LDS stages a step's DMA has one
// step (~1.1 us) to land, a miss takes longer, and every step holds a few hundred misses.  Three
// stages do not fit the LDS.  Instead each step also issues two 4-byte LDS-DMAs per thread (into
// a dump area: no VGPR is written) that touch the A and W lines of step s+3, so a missing line is
// already on its way from the Infinity Cache / HBM two steps before the real DMA asks for it.
// The touches are the youngest VMEM ops of a step: `vmcnt(2)` waits for the stage, not for them.
#include "gemm_core.h"
#include "gemm_core_il.h"
#include "gemm_epilogue.h"
#include "lab.h"
#include "gemm_core_lab.h"

namespace vr {

constexpr int G256T_SMEM = G256_SMEM_BYTES + 256;

__device__ __forceinline__ const char* uptr(const char* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void stage256t(const char* base, const uint32_t (&off)[4], char* tile, int wave) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        uint32_t o = off[t];
        asm volatile("" : "+v"(o));
        __builtin_amdgcn_global_load_lds(VR_GLOBAL(base + o), VR_LDS(tile + (wave * 32 + t * 8) * 128), 16, 0, 0);
    }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256t_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = (p.N + G256_BN - 1) / G256_BN;
    const int tiles_m = (p.M + G256_BM - 1) / G256_BM;
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int GM = p.raster_gm > 0 ? p.raster_gm : 1;
    const int gsz = GM * tiles_n;
    const int g = t / gsz, r = t % gsz;
    const int gm = min(GM, tiles_m - g * GM);
    const int m0 = __builtin_amdgcn_readfirstlane((g * GM + r % gm) * G256_BM);
    const int n0 = __builtin_amdgcn_readfirstlane((r / gm) * G256_BN);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fq = lane >> 4;
    const int arow = wm * 128 + fr, wrow = wn * 64 + fr;
    const char* A = (const char*)p.A + (size_t)m0 * p.lda * 2;
    const char* W = (const char*)p.W + (size_t)n0 * p.ldw * 2;
    const int nk = p.K / GEMM_BK;
    char* dump = smem + G256_SMEM_BYTES;
    uint32_t offA[4], offW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
        const int kc8 = ((lane & 7) ^ (row & 7)) << 3;
        offA[i] = (uint32_t)(row * p.lda + kc8) * 2u;
        offW[i] = (uint32_t)(row * p.ldw + kc8) * 2u;
    }
    const uint32_t tA = (uint32_t)((tid >> 1) * p.lda + (tid & 1) * 32) * 2u;
    const uint32_t tW = (uint32_t)((tid >> 1) * p.ldw + (tid & 1) * 32) * 2u;
    auto touch = [&](int kt) {
        kt = min(kt, nk - 1);
        uint32_t oa = tA, ow = tW;
        asm volatile("" : "+v"(oa), "+v"(ow));
        __builtin_amdgcn_global_load_lds(VR_GLOBAL(uptr(A + kt * (GEMM_BK * 2)) + oa), VR_LDS(dump), 4, 0, 0);
        __builtin_amdgcn_global_load_lds(VR_GLOBAL(uptr(W + kt * (GEMM_BK * 2)) + ow), VR_LDS(dump), 4, 0, 0);
    };
    constexpr int SB = 2 * G256_TILE_BYTES;
    touch(1);
    touch(2);
    stage256t(A, offA, smem, wave);
    stage256t(W, offW, smem + G256_TILE_BYTES, wave);

    gemm256_acc_t acc;
    gemm256_zero(acc);
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * SB;
        char* nxt = smem + ((kt + 1) & 1) * SB;
        if (kt == 0) VR_WAIT_VM_BARRIER(0);
        else VR_WAIT_VM_BARRIER(2);
        if (kt + 1 < nk) {
            stage256t(uptr(A + (kt + 1) * (GEMM_BK * 2)), offA, nxt, wave);
            stage256t(uptr(W + (kt + 1) * (GEMM_BK * 2)), offW, nxt + G256_TILE_BYTES, wave);
        }
        touch(kt + 3);
        gemm256_compute_il(acc, cur, cur + G256_TILE_BYTES, arow, wrow, fq);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if constexpr (EPI == EPI_RESID) {
        if (!p.rowmap) { gemm_epilogue_resid_tile<8, 4, 4>(acc, p, m0 + wm * 128 + fr, n0 + wn * 64, fq); return; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
        gemm_epilogue_row<EPI>(acc[i], p, m0 + wm * 128 + i * 16 + fr, n0 + wn * 64, fq);
}

template <int EPI>
static hipError_t launch_t(GemmArgs a, hipStream_t s) {
    const int tm = (a.M + G256_BM - 1) / G256_BM, tn = (a.N + G256_BN - 1) / G256_BN;
    if (a.raster_gm <= 0) a.raster_gm = tm <= 16 ? tm : 4;
    auto k = gemm256t_bf16_kernel<EPI>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G256T_SMEM); attr = true; }
    hipLaunchKernelGGL(k, dim3(tm * tn), dim3(512), G256T_SMEM, s, a);
    return hipGetLastError();
}

hipError_t launch_gemm256t(const GemmArgs& a, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_BF16: return launch_t<EPI_BF16>(a, s);
        case EPI_GELU: return launch_t<EPI_GELU>(a, s);
        case EPI_F32: return launch_t<EPI_F32>(a, s);
        case EPI_RESID: return launch_t<EPI_RESID>(a, s);
        case EPI_SWIGLU: return launch_t<EPI_SWIGLU>(a, s);
        case EPI_ROPE: return launch_t<EPI_ROPE>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vr
