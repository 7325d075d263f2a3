// 256 x 256 x 64 bf16 GEMM tile with ONE WAVE PER SIMD: 4 waves (2 x 2), 128 x 128 outputs per wave.
//
// Why: the 8-wave 256^2 kernel (gemm.hip) reads 12 fragments per 32 MFMAs and keeps two waves per
// SIMD in lockstep behind one barrier per K-step; PMC puts its main loop at ~67 % matrix-pipe use
// with the LDS ~75 % busy.  A 128 x 128 register tile reads 16 fragments per 64 MFMAs (a third less
// LDS traffic per flop) and leaves the SIMD to a single wave whose instruction stream is laid out
// by hand, so nothing has to be hidden by a second wave:
//
//   * the 256 fp32 accumulators of a lane live in the accumulation registers (a wave alone on its
//     SIMD owns all 512 registers), the 32 operand fragments of a whole K-step in VGPRs;
//   * software pipeline over K-steps (two LDS stages of A 32 KiB + W 32 KiB):
//       phase 1: 64 MFMAs on the k-half-0 fragments; under them the 16 fragment reads of k-half 1,
//                then `lgkmcnt(0) + barrier` (every wave has the whole stage in registers: the stage
//                is free) and the first LDS-DMA loads of K-step kt+2 INTO THE STAGE BEING COMPUTED;
//       phase 2: 64 MFMAs on the k-half-1 fragments; under them `vmcnt(5) + barrier` (K-step kt+1
//                has landed; the loads of kt+2 issued so far stay in flight), the rest of the loads
//                of kt+2 and the 16 k-half-0 fragment reads of K-step kt+1.
//     Loads run two K-steps ahead with two LDS stages because the registers hold the step being
//     computed; the MFMA stream never stops at a step boundary.
//   * the 16 LDS-DMA loads a wave issues per K-step are SPREAD, one per five MFMAs.  In-kernel clock
//     counts per K-step (2048 = the 128 MFMAs alone): MFMAs + barriers 2100, + fragment reads 2094,
//     + the 16 loads issued back to back 2777, + the same loads spread 2201.  The CU moves one 1 KiB
//     LDS-DMA instruction per ~16 clocks; four waves issuing theirs in one burst queue behind each
//     other and, being alone on their SIMDs, stall their MFMA streams while they wait.
//   * LDS-DMA through a buffer descriptor: per-lane offset in a VGPR (constant + K offset: two VALU
//     adds per K-step), row-group offsets in SGPRs, the two K-steps past the end of K are pointed
//     out of the descriptor's range (zeros written to the dead stage, no memory traffic), so every
//     K-step runs the same code and the `vmcnt(5)` count always holds.
//
//   * the epilogue stages through 16 KiB of LDS above the two stages, so it does not wait for the last
//     (out-of-range) loads to land.  A PERSISTENT form of this kernel (one workgroup per CU walking its
//     XCD's tile range, the load pipeline running on across tile boundaries: no prologue, no workgroup
//     launch between tiles) was built and measured in round 2 — tile period 28.7 instead of 30.1 us, but
//     the kernel no faster: with a static split the slowest workgroup sets the end (CUs differ by 8 %,
//     XCDs by 2.5 %), and per-XCD atomic cursors cost as much (index arithmetic and the LDS mailbox between
//     K-steps) as they balance: ViT qkv 227 vs 221 us, proj 109 vs 102.  The hardware dispatcher already
//     does that balancing for a one-tile-per-workgroup grid.  Round 5 tried the middle: PAIRS of tiles per workgroup for
//     whole rounds of the CUs, the rest one tile each — correct, qkv -1.7 %, but the step +1.6 % (the residual GEMMs' first
//     epilogue queues behind the second tile's HBM operand loads; and a prologue is a wait, not work: under the power cap the
//     clock takes such cycles back — DESIGN.md 5.2, ledger 50).  What did pay in round 5 was executing less in the epilogues
//     (buffer descriptors instead of predicates, branches and 64-bit address arithmetic: gemm256w_kernel.h, gemm_epilogue.h).
//
// NJ = 16-column fragments per wave: 8 -> 256 x 256 tile, 6 -> 256 x 192 tile (N = 1152 = 6 x 192: SigLIP proj / fc2,
// residual epilogue only).  Slot s = strip * NJ + fragment numbers the MFMAs of a phase; the numbers in the
// schedule above are those of NJ = 8, the NJ = 6 ones are scaled (see the constants in the kernel).
//
// LDS image, swizzle and epilogues are those of the 8-wave kernel (gemm_core.h, gemm_epilogue.h);
// the accumulators are kept as two 64-column halves so the epilogue helpers apply unchanged.
// Roofline: MFMA (2*M*N*K flops per launch).
#include <type_traits>

#include "gemm_core.h"
#include "gemm_epilogue.h"
#include "gemm256w_acc.h"

namespace vr {

bool gemm256w_fits(const GemmArgs& a, int bn);

namespace {

constexpr int W_STAGE = 2 * G256_TILE_BYTES;       // A tile + W tile = 64 KiB
constexpr unsigned W_OOB = 0x80000000u;            // per-lane offset beyond the descriptor's range

}  // namespace

// PLAIN: the host has checked that the bf16-output epilogue needs no per-row lookups (no row map / row bias,
// N and ldo multiples of 8) — the kernel then carries only the lookup-free staged epilogue.
constexpr int W_STAGING = 4096;                            // epilogue staging per wave: one piece of 32 rows x 64 bf16
constexpr int W_SMEM_BYTES = 2 * W_STAGE + 4 * W_STAGING;  // two stages + staging = 144 KiB

#define W_KERNEL_TEMPLATE template <int EPI, bool PLAIN, int NJ>
#define W_KERNEL_NAME gemm256w_bf16_kernel
#include "gemm256w_kernel.h"
#undef W_KERNEL_TEMPLATE
#undef W_KERNEL_NAME

template <int EPI, bool PLAIN, int NJ>
static hipError_t launch_wp(GemmArgs a, hipStream_t s) {
    constexpr int BN = 32 * NJ;
    const int tn = (a.N + BN - 1) / BN, tm = (a.M + G256_BM - 1) / G256_BM;
    if (a.raster_gm <= 0) a.raster_gm = tm <= 16 ? tm : 4;
    if (!gemm256w_fits(a, BN)) return hipErrorInvalidValue;
    int total = tn * tm;
    if (a.ksplit > 1) {
        if (EPI != EPI_F32 || a.K % (a.ksplit * GEMM_BK) || a.rowmap || a.rowbias) return hipErrorInvalidValue;
        total *= a.ksplit;
    }
    auto k = gemm256w_bf16_kernel<EPI, PLAIN, NJ>;
    static unsigned long long attr = 0;     // bit d: set on device d
    set_max_dynamic_lds((const void*)k, W_SMEM_BYTES, attr);
    hipLaunchKernelGGL(k, dim3(total), dim3(256), W_SMEM_BYTES, s, a);
    return hipGetLastError();
}

template <int EPI>
static hipError_t launch_w(const GemmArgs& a, hipStream_t s) {
    if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU || EPI == EPI_SWIGLU || EPI == EPI_ROPE) {
        // (a row bias rides the lookup-free bf16 form when no tile can wrap around its period and it covers whole 64-column blocks:
        // the resampler's k | v in-projection, pos_k[row % 1024] on the k half)
        const bool rb_ok = !a.rowbias || (EPI == EPI_BF16 && a.rowbias_period > 0 && a.rowbias_period % G256_BM == 0 && a.rowbias_cols % 64 == 0 && a.rowbias_cols <= a.rowbias_ld &&
                                          (size_t)a.rowbias_period * a.rowbias_ld * 4 < (1ull << 32));
        if (!a.rowmap && rb_ok && (a.N & 7) == 0 && (a.ldo & 7) == 0) return launch_wp<EPI, true, 8>(a, s);
    }
    return launch_wp<EPI, false, 8>(a, s);
}

// the LDS-DMA addresses whole matrices through 32-bit offsets: both operands (rows padded to the tile) below 2 GiB
bool gemm256w_fits(const GemmArgs& a, int bn) {
    const size_t tm = (a.M + G256_BM - 1) / G256_BM, tn = (a.N + bn - 1) / bn;
    // ... and the buffer-descriptor epilogues count a tile's output rows (and the row-bias table) in 32-bit byte offsets whose
    // out-of-range marker is +2^31: 256 rows of fp32 at pitch ldo must stay below that
    return tm * G256_BM * (size_t)a.lda * 2 < (1ull << 31) && tn * bn * (size_t)a.ldw * 2 < (1ull << 31) &&
           (size_t)G256_BM * (size_t)a.ldo * 4 < (1ull << 31) && (!a.rowbias || (size_t)G256_BM * (size_t)a.rowbias_ld * 4 < (1ull << 31));
}

hipError_t launch_gemm256w(const GemmArgs& a, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_BF16: return launch_w<EPI_BF16>(a, s);
        case EPI_GELU: return launch_w<EPI_GELU>(a, s);
        case EPI_F32: return launch_w<EPI_F32>(a, s);
        case EPI_RESID: return launch_w<EPI_RESID>(a, s);
        case EPI_SWIGLU: return launch_w<EPI_SWIGLU>(a, s);
        case EPI_ROPE: return launch_w<EPI_ROPE>(a, s);
        default: return hipErrorInvalidValue;
    }
}

// 256 x 192 tile: N % 192 == 0, residual (the SigLIP proj / fc2 GEMMs) and fp32 (split-K partial products of the
// decoder's o / down projections) epilogues; W rows readable up to the next multiple of 192 (N itself)
hipError_t launch_gemm192w(const GemmArgs& a, int epi, hipStream_t s) {
    if (a.N % 192) return hipErrorInvalidValue;
    if (epi == EPI_RESID && a.ksplit <= 1) return launch_wp<EPI_RESID, false, 6>(a, s);
    if (epi == EPI_F32) return launch_wp<EPI_F32, false, 6>(a, s);          // incl. split-K partial products
    return hipErrorInvalidValue;
}

}  // namespace vr
