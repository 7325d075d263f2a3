// 256 x 256 x 64 tile, 2 LDS stages, fragment reads INTERLEAVED with the MFMAs.
//
// gemm256_compute_tile issues its 12 fragment reads, waits for all of them (lgkmcnt(0)) and only
// then starts 32 MFMAs — twice per K-tile, on both waves of a SIMD at the same time (the barrier
// keeps them in lockstep), so the matrix pipe idles for one LDS round trip per half tile.
// Here the 32 MFMAs of a K-half run as four groups of 8 (two 16-row strips x four column
// fragments) and the two A fragments of the NEXT group (or, in the last group, the six fragments
// that open the next K-half) are requested before the current group's MFMAs are issued: the reads
// fly under ~140 cycles of matrix work, the wave never waits for more than its first six reads per
// K-tile.  sched_barrier(0) pins the hand-written order; the compiler still inserts the counted
// lgkmcnt waits.  (The first reads of a K-tile are issued right after the stage barrier: the
// previous tile's MFMAs have drained by then, their latency is the one exposed LDS round trip.)
#pragma once
#include "gemm_core.h"

namespace vr {

__device__ __forceinline__ bf16x8 g256_frag(const char* t, int row, int kk, int fq) {
    return *reinterpret_cast<const bf16x8*>(t + row * 128 + (((kk * 4 + fq) ^ (row & 7)) << 4));
}

// the MFMAs of ONE K-step (64 wide) from a landed stage.  Two register sets for the A fragments (group G reads
// set G & 1 while set (G + 1) & 1 is being filled) and one W set per K-half: every index is a compile-time
// constant after unrolling, so there are NO register copies (the rotating a0 = b0 form of round 1 cost 128
// v_mov per K-step and wave — a quarter of the loop's issue slots).
__device__ __forceinline__ void gemm256_compute_il(gemm256_acc_t& acc, const char* tA, const char* tW, int arow,
                                                   int wrow, int fq) {
    bf16x8 w[2][4], a[2][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[0][j] = g256_frag(tW, wrow + j * 16, 0, fq);
    a[0][0] = g256_frag(tA, arow, 0, fq);
    a[0][1] = g256_frag(tA, arow + 16, 0, fq);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cur = (kk * 4 + g) & 1, nxt = cur ^ 1;
            // ---- request what the NEXT group needs
            if (g < 3) {
                a[nxt][0] = g256_frag(tA, arow + (2 * g + 2) * 16, kk, fq);
                a[nxt][1] = g256_frag(tA, arow + (2 * g + 3) * 16, kk, fq);
            } else if (kk == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) w[1][j] = g256_frag(tW, wrow + j * 16, 1, fq);
                a[nxt][0] = g256_frag(tA, arow, 1, fq);
                a[nxt][1] = g256_frag(tA, arow + 16, 1, fq);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- 8 MFMAs of this group: strips 2g, 2g+1
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[2 * g][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[kk][j], a[cur][0], acc[2 * g][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[2 * g + 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[kk][j], a[cur][1], acc[2 * g + 1][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__device__ __forceinline__ void gemm256_mainloop_il(gemm256_acc_t& acc, const bf16_t* __restrict__ A, int lda,
                                                    const bf16_t* __restrict__ W, int ldw, int m0, int n0,
                                                    int K, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int arow = wm * 128 + fr, wrow = wn * 64 + fr;
    const int nk = K / GEMM_BK;
    stage_glds(A, lda, m0, 0, smem, wave, lane);
    stage_glds(W, ldw, n0, 0, smem + G256_TILE_BYTES, wave, lane);
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * 2 * G256_TILE_BYTES;
        char* nxt = smem + ((kt + 1) & 1) * 2 * G256_TILE_BYTES;
        __syncthreads();
        if (kt + 1 < nk) {
            stage_glds(A, lda, m0, (kt + 1) * GEMM_BK, nxt, wave, lane);
            stage_glds(W, ldw, n0, (kt + 1) * GEMM_BK, nxt + G256_TILE_BYTES, wave, lane);
        }
        gemm256_compute_il(acc, cur, cur + G256_TILE_BYTES, arow, wrow, fq);
    }
    __syncthreads();
}

}  // namespace vr
