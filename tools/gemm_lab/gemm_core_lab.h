// Measured-negative main-loop experiments of the 256 x 256 tile (kept out of the product library;
// tools/gemm_lab/build.py): register staging of the 128^2 tile, the plain 2-stage 256^2 loop and
// the BK = 32 x 4-stage loop with counted vmcnt.
#pragma once
#include "gemm_core.h"

namespace vr {

// ---- register staging ------------------------------------------------------------------
__device__ __forceinline__ void stage_load(const bf16_t* __restrict__ g, int ld, int row0, int k0,
                                           int tid, u32x4 (&regs)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = tid + 256 * t;
        const int row = c >> 3, kc = c & 7;
        regs[t] = *reinterpret_cast<const u32x4*>(g + (size_t)(row0 + row) * ld + k0 + kc * 8);
    }
}
__device__ __forceinline__ void stage_write(char* tile, int tid, const u32x4 (&regs)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = tid + 256 * t;
        const int row = c >> 3, kc = c & 7;
        *reinterpret_cast<u32x4*>(tile + row * 128 + ((kc ^ (row & 7)) << 4)) = regs[t];
    }
}

// 128^2 tile with register staging (issue early, write late)
__device__ __forceinline__ void gemm_mainloop_reg(gemm_acc_t& acc, const bf16_t* __restrict__ A, int lda,
                                                  const bf16_t* __restrict__ W, int ldw, int m0, int n0,
                                                  int K, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = K / GEMM_BK;
    u32x4 ra[4], rw[4];
    stage_load(A, lda, m0, 0, tid, ra);
    stage_load(W, ldw, n0, 0, tid, rw);
    stage_write(smem, tid, ra);
    stage_write(smem + GEMM_TILE_BYTES, tid, rw);
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * 2 * GEMM_TILE_BYTES;
        char* nxt = smem + ((kt + 1) & 1) * 2 * GEMM_TILE_BYTES;
        __syncthreads();
        if (kt + 1 < nk) {
            stage_load(A, lda, m0, (kt + 1) * GEMM_BK, tid, ra);
            stage_load(W, ldw, n0, (kt + 1) * GEMM_BK, tid, rw);
        }
        gemm_compute_tile(acc, cur, cur + GEMM_TILE_BYTES, wm, wn, lane);
        if (kt + 1 < nk) {
            stage_write(nxt, tid, ra);
            stage_write(nxt + GEMM_TILE_BYTES, tid, rw);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void gemm256_compute_tile(gemm256_acc_t& acc, const char* tA, const char* tW,
                                                     int wm, int wn, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        bf16x8 a[8], w[4];
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
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], a[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
}

__device__ __forceinline__ void gemm256_mainloop(gemm256_acc_t& acc, const bf16_t* __restrict__ A, int lda,
                                                 const bf16_t* __restrict__ W, int ldw, int m0, int n0,
                                                 int K, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = K / GEMM_BK;
    stage_glds(A, lda, m0, 0, smem, wave, lane);
    stage_glds(W, ldw, n0, 0, smem + G256_TILE_BYTES, wave, lane);
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * 2 * G256_TILE_BYTES;
        char* nxt = smem + ((kt + 1) & 1) * 2 * G256_TILE_BYTES;
        __syncthreads();   // tile kt landed (LDS-DMA drained) and tile kt-1 fully consumed
        if (kt + 1 < nk) {
            stage_glds(A, lda, m0, (kt + 1) * GEMM_BK, nxt, wave, lane);
            stage_glds(W, ldw, n0, (kt + 1) * GEMM_BK, nxt + G256_TILE_BYTES, wave, lane);
        }
        gemm256_compute_tile(acc, cur, cur + G256_TILE_BYTES, wm, wn, lane);
    }
    __syncthreads();
}

}  // namespace vr

// =============================================================================================
// 256 x 256 tile, deep pipeline: K-step 32, FOUR LDS stages (4 x 32 KiB), LDS-DMA issued THREE
// steps ahead.  The barrier of step t waits only for the loads of step t (counted
// `s_waitcnt vmcnt(8)`: the 2x4 loads of steps t+1, t+2 stay in flight across it) — the 2-stage
// loop above drains vmcnt(0) every step and was bound by the ~1 us L2/HBM latency of loads
// issued just one step earlier.  Raw s_barrier + inline waitcnt: __syncthreads() would make
// hipcc drain the LDS-DMA queue.
// LDS image: 256 rows x 64 B per operand; 16-B chunk c of row r at chunk c ^ swz[(r>>2)&3],
// swz = {0,2,3,1}: conflict-free for the ds_read_b128 lane groups.
// =============================================================================================
namespace vr {

constexpr int P4_BK = 32;
constexpr int P4_TILE_BYTES = 256 * P4_BK * 2;              // 16 KiB per operand per stage
constexpr int P4_STAGE_BYTES = 2 * P4_TILE_BYTES;           // 32 KiB
constexpr int P4_SMEM_BYTES = 4 * P4_STAGE_BYTES;           // 128 KiB

__device__ __forceinline__ int p4_swz(int row) { return (0x1320 >> (((row >> 2) & 3) << 2)) & 3; }

// this wave fills rows [32*wave, 32*wave+32) of one operand stage: 2 x (16 rows x 64 B)
__device__ __forceinline__ void p4_stage(const bf16_t* __restrict__ g, int ld, int row0, int k0, char* tile,
                                         int wave, int lane) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int rbase = wave * 32 + t * 16;
        const int row = rbase + (lane >> 2);
        const int kc = (lane & 3) ^ p4_swz(row);
        const bf16_t* src = g + (size_t)(row0 + row) * ld + k0 + kc * 8;
        __builtin_amdgcn_global_load_lds(VR_GLOBAL(src), VR_LDS(tile + rbase * 64), 16, 0, 0);
    }
}

__device__ __forceinline__ void p4_compute(gemm256_acc_t& acc, const char* tA, const char* tW, int wm, int wn,
                                           int lane) {
    const int fr = lane & 15, fq = lane >> 4;
    bf16x8 a[8], w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + fr;
        w[j] = *reinterpret_cast<const bf16x8*>(tW + row * 64 + ((fq ^ p4_swz(row)) << 4));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wm * 128 + i * 16 + fr;
        a[i] = *reinterpret_cast<const bf16x8*>(tA + row * 64 + ((fq ^ p4_swz(row)) << 4));
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], a[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
}


__device__ __forceinline__ void gemm256_mainloop_p4(gemm256_acc_t& acc, const bf16_t* __restrict__ A, int lda,
                                                    const bf16_t* __restrict__ W, int ldw, int m0, int n0,
                                                    int K, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = K / P4_BK;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s < nk) {
            p4_stage(A, lda, m0, s * P4_BK, smem + s * P4_STAGE_BYTES, wave, lane);
            p4_stage(W, ldw, n0, s * P4_BK, smem + s * P4_STAGE_BYTES + P4_TILE_BYTES, wave, lane);
        }
    }
    for (int kt = 0; kt < nk; ++kt) {
        // step kt's own 4 loads have landed when at most the 4/8 younger ones are outstanding;
        // the barrier then (a) publishes every wave's part of stage kt, (b) proves stage kt-1 is
        // no longer being read, so it can be refilled with step kt+3 below.
        if (kt + 2 < nk) VR_WAIT_VM_BARRIER(8);
        else if (kt + 1 < nk) VR_WAIT_VM_BARRIER(4);
        else VR_WAIT_VM_BARRIER(0);
        if (kt + 3 < nk) {
            char* nxt = smem + ((kt + 3) & 3) * P4_STAGE_BYTES;
            p4_stage(A, lda, m0, (kt + 3) * P4_BK, nxt, wave, lane);
            p4_stage(W, ldw, n0, (kt + 3) * P4_BK, nxt + P4_TILE_BYTES, wave, lane);
        }
        const char* cur = smem + (kt & 3) * P4_STAGE_BYTES;
        p4_compute(acc, cur, cur + P4_TILE_BYTES, wm, wn, lane);
    }
    asm volatile("s_barrier" ::: "memory");
}

}  // namespace vr
