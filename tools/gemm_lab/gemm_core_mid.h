// 256 x 256 x 64 tile, two LDS stages, LDS-DMA issued in the MIDDLE of the previous K-tile.
//
// The plain 2-stage loop (gemm256_mainloop) issues the loads of tile t+1 after the barrier that
// opens tile t and drains them (vmcnt(0)) at the barrier that opens tile t+1: one tile of
// look-ahead, and the measured iteration time (~1.96 us) is the LDS-DMA latency, not the MFMA time
// (~1.0 us).  Here a second barrier sits after the LAST fragment reads of tile t (before its last
// 32 MFMAs); past it nobody reads stage t&1 any more, so the loads of tile t+2 go out right there
// and have the rest of tile t plus all of tile t+1 to land: 1.5 tiles of look-ahead with the same
// 128 KiB of LDS.  Waits are counted (`vmcnt(8)` leaves the 8 loads of the younger tile in flight
// across the barrier); raw s_barrier, because __syncthreads() would drain the LDS-DMA queue.
#pragma once
#include "gemm_core.h"

namespace vr {

__device__ __forceinline__ void g256_read_frags(bf16x8 (&a)[8], bf16x8 (&w)[4], const char* tA, const char* tW,
                                                int wm, int wn, int lane, int kk) {
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + fr;
        w[j] = *reinterpret_cast<const bf16x8*>(tW + row * 128 + (((kk * 4 + fq) ^ (row & 7)) << 4));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wm * 128 + i * 16 + fr;
        a[i] = *reinterpret_cast<const bf16x8*>(tA + row * 128 + (((kk * 4 + fq) ^ (row & 7)) << 4));
    }
}

__device__ __forceinline__ void g256_mfma(gemm256_acc_t& acc, const bf16x8 (&a)[8], const bf16x8 (&w)[4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], a[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
}

__device__ __forceinline__ void gemm256_mainloop_mid(gemm256_acc_t& acc, const bf16_t* __restrict__ A, int lda,
                                                     const bf16_t* __restrict__ W, int ldw, int m0, int n0,
                                                     int K, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = K / GEMM_BK;
    stage_glds(A, lda, m0, 0, smem, wave, lane);
    stage_glds(W, ldw, n0, 0, smem + G256_TILE_BYTES, wave, lane);
    if (nk > 1) {
        stage_glds(A, lda, m0, GEMM_BK, smem + 2 * G256_TILE_BYTES, wave, lane);
        stage_glds(W, ldw, n0, GEMM_BK, smem + 3 * G256_TILE_BYTES, wave, lane);
    }
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * 2 * G256_TILE_BYTES;
        // B1: tile kt has landed in every wave's part of the stage (tile kt+1 may still fly)
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        bf16x8 a[8], w[4];
        g256_read_frags(a, w, cur, cur + G256_TILE_BYTES, wm, wn, lane, 0);
        g256_mfma(acc, a, w);
        g256_read_frags(a, w, cur, cur + G256_TILE_BYTES, wm, wn, lane, 1);
        // B2: every wave holds its last fragments of this stage in registers -> refill it
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 2 < nk) {
            stage_glds(A, lda, m0, (kt + 2) * GEMM_BK, cur, wave, lane);
            stage_glds(W, ldw, n0, (kt + 2) * GEMM_BK, cur + G256_TILE_BYTES, wave, lane);
        }
        g256_mfma(acc, a, w);
    }
    asm volatile("s_barrier" ::: "memory");
}

}  // namespace vr
