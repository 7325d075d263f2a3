// bf16 MFMA GEMM main loop for gfx950, shared by the dense GEMM kernels (gemm.hip) and the
// fused similarity + top-k search kernel (search.hip).
//
//   acc[i][j] (+)= A[m0.., :K] * W[n0.., :K]^T      A [M][lda], W [N][ldw] bf16, K-contiguous
//
// Block tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves in a 2x2 grid, each wave owns a
// 64x64 sub-tile as 4x4 fragments of v_mfma_f32_16x16x32_bf16 (64 fp32 accumulators / lane).
// The W fragment is fed as the MFMA "A" operand and the activation fragment as "B", so a
// lane ends up with FOUR CONSECUTIVE OUTPUT COLUMNS of one output row:
//     acc[i][j][r]  =  out[ m0 + wm*64 + i*16 + (lane&15) ][ n0 + wn*64 + j*16 + (lane>>4)*4 + r ]
// which makes bias loads and bf16/f32 stores 8/16-byte vector accesses.
//
// LDS image of one operand tile: 128 rows x 128 B (64 bf16), 16-byte chunk c of row r stored
// at chunk (c ^ (r & 7)) — the XOR swizzle that makes the ds_read_b128 fragment reads
// conflict-free (cdna_hip_programming.md T2).  Staging is global_load_lds_dwordx4 (LDS-DMA, no
// VGPR round trip): the LDS destination of a wave instruction is lane-linear, so the swizzle is
// applied to the per-lane SOURCE address (rule 21).
// Double-buffered: ONE barrier per K-tile; tile t+1 is in flight while tile t is computed.
#pragma once
#include "common.h"

namespace vr {

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 64;
constexpr int GEMM_TILE_BYTES = GEMM_BM * GEMM_BK * 2;      // 16 KiB per operand tile
constexpr int GEMM_SMEM_BYTES = 4 * GEMM_TILE_BYTES;        // 2 buffers x (A + W) = 64 KiB

typedef f32x4 gemm_acc_t[4][4];

__device__ __forceinline__ void gemm_zero(gemm_acc_t& acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ---- LDS-DMA staging: this wave fills rows [wave*32, wave*32+32) of one operand tile ----
__device__ __forceinline__ void stage_glds(const bf16_t* __restrict__ g, int ld, int row0, int k0,
                                           char* tile, int wave, int lane) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int rbase = wave * 32 + t * 8;
        const int row = rbase + (lane >> 3);
        const int kc = (lane & 7) ^ (row & 7);
        const bf16_t* src = g + (size_t)(row0 + row) * ld + k0 + kc * 8;
        __builtin_amdgcn_global_load_lds(VR_GLOBAL(src), VR_LDS(tile + rbase * 128), 16, 0, 0);
    }
}

// ---- one K-tile of MFMAs from LDS --------------------------------------------------------
__device__ __forceinline__ void gemm_compute_tile(gemm_acc_t& acc, const char* tA, const char* tW,
                                                  int wm, int wn, int lane) {
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        bf16x8 a[4], w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wm * 64 + i * 16 + fr;
            a[i] = *reinterpret_cast<const bf16x8*>(tA + row * 128 + (((kk * 4 + fq) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wn * 64 + j * 16 + fr;
            w[j] = *reinterpret_cast<const bf16x8*>(tW + row * 128 + (((kk * 4 + fq) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j], a[i], acc[i][j], 0, 0, 0);
    }
}

// Accumulate the full K range of one 128x128 output tile.  smem: GEMM_SMEM_BYTES, 16-B aligned.
// Rows [m0, m0+128) of A and [n0, n0+128) of W must be readable (buffers are padded).
__device__ __forceinline__ void gemm_mainloop(gemm_acc_t& acc, const bf16_t* __restrict__ A, int lda,
                                              const bf16_t* __restrict__ W, int ldw, int m0, int n0,
                                              int K, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = K / GEMM_BK;
    stage_glds(A, lda, m0, 0, smem, wave, lane);
    stage_glds(W, ldw, n0, 0, smem + GEMM_TILE_BYTES, wave, lane);
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * 2 * GEMM_TILE_BYTES;
        char* nxt = smem + ((kt + 1) & 1) * 2 * GEMM_TILE_BYTES;
        __syncthreads();   // tile kt landed (LDS-DMA drained) and tile kt-1 fully consumed
        if (kt + 1 < nk) {
            stage_glds(A, lda, m0, (kt + 1) * GEMM_BK, nxt, wave, lane);
            stage_glds(W, ldw, n0, (kt + 1) * GEMM_BK, nxt + GEMM_TILE_BYTES, wave, lane);
        }
        gemm_compute_tile(acc, cur, cur + GEMM_TILE_BYTES, wm, wn, lane);
    }
    __syncthreads();   // smem may be reused by the caller (epilogue staging)
}

}  // namespace vr

// =============================================================================================
// 256 x 256 x 64 tile, 512 threads = 8 waves as 2 (M) x 4 (N); each wave owns 128 x 64 = 8 x 4
// fragments (128 fp32 accumulators).  Per MFMA this halves the LDS fragment reads (12 b128 reads
// feed 32 MFMAs instead of 8 feeding 16) and the LDS-DMA bytes of the 128x128 kernel, which is
// what bounds that kernel (MfmaUtil 43 %, no bank conflicts, waves parked on LDS/barrier).
// LDS: 2 stages x (A 32 KiB + W 32 KiB) = 128 KiB -> one workgroup per CU, 2 waves per SIMD.
// Same LDS image / swizzle as above; wave w stages rows [32w, 32w+32) of both operands.
// =============================================================================================
namespace vr {

constexpr int G256_BM = 256, G256_BN = 256;
constexpr int G256_TILE_BYTES = 256 * GEMM_BK * 2;          // 32 KiB per operand tile
constexpr int G256_SMEM_BYTES = 4 * G256_TILE_BYTES;        // 128 KiB

typedef f32x4 gemm256_acc_t[8][4];

__device__ __forceinline__ void gemm256_zero(gemm256_acc_t& acc) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

}  // namespace vr
