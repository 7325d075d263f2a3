// 256 x 256 x 64 tile, two LDS stages, STAGGERED wave groups ("ping-pong").
//
// PMC of the plain 2-stage loop (8192^3): MFMA pipe 51 % busy, LDS 32 %, no bank conflicts; waves
// 37 % parked in s_waitcnt/s_barrier.  Cause: the barrier keeps all 8 waves in lockstep, so the two
// waves that share a SIMD (wave w and w+4) read their fragments at the same time and then fight
// for the matrix pipe at the same time — the ~500-cycle LDS read latency is exposed twice per
// K-tile on every SIMD.
//
// Here the workgroup runs as two groups of four waves (group = wave >> 2 = the M half they own;
// one wave of each group per SIMD) executing the SAME per-tile sequence
//     R0 | M0 | R1 | M1 |        R = 12 ds_read_b128 (+ wait), M = 32 MFMA, | = s_barrier
// but group 1 starts one barrier interval late: while one wave of a SIMD issues its 32 MFMAs the
// other one has its fragment reads in flight, and vice versa.  Every interval costs
// max(LDS latency, 32 MFMA) instead of their sum.
// LDS-DMA for tile t+1 is issued at global interval 4t (group 0: start of R0(t); group 1: start of
// M1(t-1)) and waited for (vmcnt(0)) at the end of interval 4t+3: four intervals of look-ahead.
#pragma once
#include "gemm_core_mid.h"

namespace vr {

#define VR_BARRIER() asm volatile("s_barrier" ::: "memory")
#define VR_LGKM_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define VR_LGKM_VM_BARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define VR_VM_BARRIER() asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ void gemm256_mainloop_stag(gemm256_acc_t& acc, const bf16_t* __restrict__ A, int lda,
                                                      const bf16_t* __restrict__ W, int ldw, int m0, int n0,
                                                      int K, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;          // wm doubles as the stagger group
    const int nk = K / GEMM_BK;
    const int SB = 2 * G256_TILE_BYTES;               // bytes per stage

    stage_glds(A, lda, m0, 0, smem, wave, lane);
    stage_glds(W, ldw, n0, 0, smem + G256_TILE_BYTES, wave, lane);
    VR_VM_BARRIER();                                  // tile 0 visible to everyone
    if (wm == 1) {                                    // interval 0: group 1 only prefetches
        if (nk > 1) {
            stage_glds(A, lda, m0, GEMM_BK, smem + SB, wave, lane);
            stage_glds(W, ldw, n0, GEMM_BK, smem + SB + G256_TILE_BYTES, wave, lane);
        }
        VR_BARRIER();
    }
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * SB;
        char* nxt = smem + ((kt + 1) & 1) * SB;
        bf16x8 a[8], w[4];
        // ---- R0
        if (wm == 0 && kt + 1 < nk) {
            stage_glds(A, lda, m0, (kt + 1) * GEMM_BK, nxt, wave, lane);
            stage_glds(W, ldw, n0, (kt + 1) * GEMM_BK, nxt + G256_TILE_BYTES, wave, lane);
        }
        g256_read_frags(a, w, cur, cur + G256_TILE_BYTES, wm, wn, lane, 0);
        VR_LGKM_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        // ---- M0
        g256_mfma(acc, a, w);
        __builtin_amdgcn_sched_barrier(0);
        VR_BARRIER();
        // ---- R1 (group 1: this is the end of global interval 4kt+3 -> tile kt+1 must have landed)
        g256_read_frags(a, w, cur, cur + G256_TILE_BYTES, wm, wn, lane, 1);
        if (wm == 1) VR_LGKM_VM_BARRIER(); else VR_LGKM_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        // ---- M1 (group 1: global interval 4(kt+1): `cur` is free, prefetch tile kt+2 into it)
        if (wm == 1 && kt + 2 < nk) {
            stage_glds(A, lda, m0, (kt + 2) * GEMM_BK, cur, wave, lane);
            stage_glds(W, ldw, n0, (kt + 2) * GEMM_BK, cur + G256_TILE_BYTES, wave, lane);
        }
        g256_mfma(acc, a, w);
        __builtin_amdgcn_sched_barrier(0);
        if (wm == 0) VR_VM_BARRIER(); else VR_BARRIER();
    }
    if (wm == 0) VR_BARRIER();                        // balances group 1's interval-0 barrier
}

}  // namespace vr
