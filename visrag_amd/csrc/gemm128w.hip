// 128 x 192 (NJ = 6) / 128 x 256 (NJ = 8) x 64 bf16 GEMM tile with ONE WAVE PER SIMD and THREE LDS stages: the
// few-rows form of gemm256w.hip (round 6).
//
// Why: the decoder's o / down projections have T = 2176 rows and N = 2304 columns — 108 tiles of 256 x 192, so rounds 2-5
// split K over two workgroups per tile and wrote fp32 partial PLANES that the RMSNorm behind them summed (88 MB per norm
// launch for a 20 MB residual stream).  Half-height tiles give 17 x 12 = 204 workgroups with the FULL K each: the residual
// epilogue adds in place, no planes exist, the norm reads the stream once.
//
// What changes against the 256-row kernel:
//   * 4 waves (2 x 2), 64 x 16 NJ outputs per wave: 4 + NJ fragment reads feed 4 NJ MFMAs per k-half (10 per 24 / 12 per 32
//     against 14 per 48 / 16 per 64) and every K-step moves 40 / 48 KiB by LDS-DMA under 48 / 64 MFMAs per wave — one 1 KiB
//     load per 4 - 5 MFMAs, close to the one-per-16-clocks the CU sustains.  With two stages the loads of step kt+2 could
//     only go out after the phase-1 barrier of step kt (the stage being computed must be in registers first): a burst.
//     A stage is 40 / 48 KiB here, so FOUR / THREE fit (the text below is the three-stage form; with four the loads run three
//     steps ahead): the loads of step kt+2 go into the stage step kt-1 used, which the one
//     barrier of step kt-1 has already released — they are spread over the WHOLE K-step, and the phase-1 barrier is gone:
//       phase 1: 4 NJ MFMAs on k-half 0; under them the fragment reads of k-half 1 (even slots) and loads (every DS slots);
//       phase 2: 4 NJ MFMAs on k-half 1; slot 1: `vmcnt(D1) lgkmcnt(0)` + barrier (K-step kt+1 has landed everywhere and
//                every wave holds the whole of stage kt in registers), the rest of the loads, the k-half-0 reads of kt+1.
//   * accumulators: the first 32 of gemm256w_acc.h's hand-allocated set (a[0:127]); build.py checks the listing likewise.
//   * epilogues: the buffer-descriptor forms of gemm256w_kernel.h on 4 pieces of 32 rows per wave — residual in place
//     (any NJ), and the lookup-free bf16 / GELU / SwiGLU forms for NJ = 8.  Shapes they do not cover (a column edge, row
//     maps, row bias, split K) are refused by the launcher: callers keep the 256-row kernels for those.
//
// Measured (round 6, DESIGN.md ledger 53-57): K-step 880 clocks warm (768 = the MFMAs alone) with four stages, 930 with three;
// in the model the o / down launches take 37 / 72 us against 32 / 58 us for the split-K pair they replace (cold weights: a
// 128-row tile consumes 40 KiB per 0.47 us K-step and HBM answers in ~2 us — more than the LDS can have in flight), the RMSNorm
// behind them 11.5 instead of 20 us: the step is level, 4.6 GB of plane traffic per step are gone.  Tried on top and dropped:
// a fifth wave per workgroup touching the operand lines eight K-steps ahead (two waves on one SIMD: K-step 1500 clocks),
// K-blocked weights (-1.7 / -2.8 us cold), prefetch workgroups on the round's idle CUs streaming the next projection's weights
// through the memory-side cache (down -4.5 us, gate / up +2.5: level).
// Roofline: MFMA (2*M*N*K flops per launch).
#include <cstdlib>
#include <type_traits>

#include "gemm_core.h"
#include "gemm_epilogue.h"
#include "gemm256w_acc.h"

namespace vr {

namespace {

constexpr int H_BM = 128;
constexpr int H_A_BYTES = H_BM * GEMM_BK * 2;      // 16 KiB
constexpr unsigned H_OOB = 0x80000000u;            // per-lane offset beyond the descriptor's range
constexpr int H_STAGING = 4096;                    // epilogue staging per wave (bf16 forms): one piece of 32 rows x 64 bf16

template <int NJ> constexpr int h_stage_bytes() { return H_A_BYTES + 32 * NJ * GEMM_BK * 2; }
// LDS stages: the 128 x 192 residual form needs no staging area, so FOUR of its 40 KiB stages fit (loads three K-steps — 1.5 us
// — ahead: the decoder's weights come from HBM, and two steps of 0.5 us were less than that takes); 128 x 256: three of 48 KiB
template <int NJ> constexpr int h_stages() { return NJ == 6 ? 4 : 3; }
template <int EPI, int NJ> constexpr int h_smem_bytes() { return h_stages<NJ>() * h_stage_bytes<NJ>() + (EPI == EPI_RESID ? 0 : 4 * H_STAGING); }

}  // namespace

template <int EPI, int NJ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm128w_bf16_kernel(GemmArgs p) {
    static_assert(NJ == 8 || (NJ == 6 && EPI == EPI_RESID), "wave tile 64 x 128 or 64 x 96");
    constexpr int NI = 4;                       // 16-row strips per wave
    constexpr int BN = 32 * NJ;                 // tile columns
    constexpr int NS = NI * NJ;                 // MFMAs per phase (slots)
    constexpr int NR = NI + NJ;                 // fragment reads per k-half = LDS-DMA loads per K-step and wave
    constexpr int STAGE = h_stage_bytes<NJ>();
    constexpr int NST = h_stages<NJ>();         // LDS stages; loads run NST - 1 K-steps ahead
    constexpr int DS = NJ == 8 ? 5 : 4;         // one load per DS slots
    constexpr int L1 = 1;                       // first load slot of phase 1
    constexpr int D1 = (NS - 1 - L1) / DS + 1;  // loads issued in phase 1
    constexpr int SB2 = 1;                      // slot of the K-step's barrier (phase 2)
    constexpr int L2 = 3;                       // first load slot of phase 2
    constexpr int NF = NJ / 2;                  // fragments per epilogue piece: 64 or 48 columns
    static_assert(D1 < NR && L2 + DS * (NR - D1 - 1) < NS && 2 * (NR - 1) < NS && SB2 + 1 + 2 * (NR - 1) < NS && L2 > SB2, "schedule fits");
    static_assert((NST == 3 && D1 == 7) || (NST == 4 && D1 == 6 && NR == 10), "the counted waits below");
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef VR_W_TIMING                   // tile anatomy (tools/w_anatomy.py, tagged builds only)
    const unsigned long long tm0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long tm1 = 0, tm2 = 0, cy1 = 0, cy2 = 0;
#endif
    const int tiles_n = p.N / BN;
    const int tiles_m = (p.M + H_BM - 1) / H_BM;
    const int total = tiles_m * tiles_n;
    const int nk = p.K / GEMM_BK;

    const int my = xcd_remap(blockIdx.x, total);
    int m0, n0;
    {
        const int GM = p.raster_gm > 0 ? p.raster_gm : 1;
        const int gsz = GM * tiles_n;
        const int g = my / gsz, r = my % gsz;
        const int gm = min(GM, tiles_m - g * GM);
        m0 = __builtin_amdgcn_readfirstlane((g * GM + r % gm) * H_BM);
        n0 = __builtin_amdgcn_readfirstlane((r / gm) * BN);
    }

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;

    // ---- LDS-DMA addressing: wave w fills rows [32 w, 32 w + 32) of the A tile and [8 NJ w, 8 NJ (w + 1)) of the W tile,
    //      8 rows per instruction; lane l -> row l / 8, 16-byte chunk (l % 8) ^ (row % 8) of the 128-byte k-slice.
    const auto arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7FFFFFFF, 0x00020000);
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7FFFFFFF, 0x00020000);
    const unsigned lchunk = (unsigned)(((lane & 7) ^ (lane >> 3)) << 4);
    const unsigned lofA = (unsigned)(lane >> 3) * (unsigned)p.lda * 2u + lchunk;
    constexpr unsigned wstep = (unsigned)(GEMM_BK * 2);      // bytes between K-steps
    const unsigned wrow = (unsigned)p.ldw * 2u;             // bytes between W rows
    const unsigned lofW = (unsigned)(lane >> 3) * wrow + lchunk;
    const unsigned rgA = (unsigned)p.lda * 16u, rgW = wrow * 8u;                  // bytes per 8-row group
    const unsigned sA0 = (unsigned)wave * (unsigned)NI * rgA, sW0 = (unsigned)wave * (unsigned)NJ * rgW;
    char* const dmaA = smem + wave * (NI * 1024);
    char* const dmaW = smem + H_A_BYTES + wave * (NJ * 1024);
    const unsigned curA = (unsigned)m0 * (unsigned)p.lda * 2u, curW = (unsigned)n0 * wrow;

    // one of the NR loads of a K-step: d < NI -> A row group d, else W row group d - NI
    auto dma = [&](int stage, int d, unsigned vA, unsigned vW) {
        if (d < NI) __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, VR_LDS(dmaA + stage * STAGE + d * 1024), 16, vA, sA0 + d * rgA, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, VR_LDS(dmaW + stage * STAGE + (d - NI) * 1024), 16, vW, sW0 + (d - NI) * rgW, 0, 0);
    };

    // ---- fragment addressing: row = strip * 16 + fr, chunk (kk * 4 + fq) ^ (row & 7); the strip is an immediate offset
    typedef const __attribute__((address_space(3))) bf16x8* frag_p;
    frag_p pA[NST][2], pW[NST][2];
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ch = ((kk * 4 + fq) ^ (fr & 7)) << 4;
            pA[st][kk] = (frag_p)VR_LDS(smem + st * STAGE + (wm * 16 * NI + fr) * 128 + ch);
            pW[st][kk] = (frag_p)VR_LDS(smem + st * STAGE + H_A_BYTES + (wn * 16 * NJ + fr) * 128 + ch);
            asm volatile("" : "+v"(pA[st][kk]), "+v"(pW[st][kk]));
        }

    bf16x8 a0[NI], w0[NJ], a1[NI], w1[NJ];

    // ---- prologue: K-steps 0 .. NST-2 in flight, the accumulators zeroed under their latency, k-half-0 fragments of step 0
    {
#pragma unroll
        for (int st = 0; st < NST - 1; ++st) {
            const unsigned kA = st < nk ? curA + (unsigned)(st * GEMM_BK * 2) : H_OOB, kW = st < nk ? curW + (unsigned)st * wstep : H_OOB;
#pragma unroll
            for (int d = 0; d < NR; ++d) dma(st, d, lofA + kA, lofW + kW);
        }
#define H_Z(n, R, C0, C1, C2, C3) if ((n) < 8 * NI) { W_ZERO(n, R, C0, C1, C2, C3) }
        W_FOR_EACH_ACC(H_Z)
#undef H_Z
        if constexpr (NST == 3) VR_WAIT_VM_BARRIER(12); else VR_WAIT_VM_BARRIER(20);      // (NST - 2) NR: step 0 has landed
#pragma unroll
        for (int j = 0; j < NJ; ++j) w0[j] = pW[0][0][j * 128];
#pragma unroll
        for (int i = 0; i < NI; ++i) a0[i] = pA[0][0][i * 128];
    }
#ifdef VR_W_TIMING
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tm1 = __builtin_amdgcn_s_memrealtime();
    cy1 = __builtin_amdgcn_s_memtime();
#endif

    {
        auto step = [&](auto stage_c, int kt) {
            constexpr int S = decltype(stage_c)::value;
            constexpr int SN = (S + 1) % NST, SL = (S + NST - 1) % NST;      // next step's stage; the stage step kt + NST - 1 lands in
            const int k2 = kt + NST - 1;
            // (the loads of the K-steps past the end of K go nowhere: out of the descriptors' range)
            const unsigned vA = lofA + (k2 < nk ? curA + (unsigned)k2 * (GEMM_BK * 2) : H_OOB);
            const unsigned vW = lofW + (k2 < nk ? curW + (unsigned)k2 * wstep : H_OOB);
            __builtin_amdgcn_sched_barrier(0);
            auto aux1 = [&](int sl) {
                if (sl < 2 * NR && (sl & 1) == 0) {
                    const int q = sl >> 1;
                    if (q < NJ) w1[q] = pW[S][1][q * 128];
                    else a1[q - NJ] = pA[S][1][(q - NJ) * 128];
                }
                if (sl >= L1 && (sl - L1) % DS == 0) dma(SL, (sl - L1) / DS, vA, vW);                      // loads 0 .. D1-1
                __builtin_amdgcn_sched_barrier(0);
            };
            auto aux2 = [&](int sl) {
                if (sl == SB2) {
                    // in flight behind K-step kt + 1: the steps between it and the one being requested, and that one's first D1 loads
                    if constexpr (NST == 3) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
                if (sl >= L2 && (sl - L2) % DS == 0 && D1 + (sl - L2) / DS < NR) dma(SL, D1 + (sl - L2) / DS, vA, vW);   // loads D1 .. NR-1
                if (sl > SB2 && ((sl - SB2 - 1) & 1) == 0 && (sl - SB2 - 1) / 2 < NR) {
                    const int q = (sl - SB2 - 1) >> 1;
                    if (q < NJ) w0[q] = pW[SN][0][q * 128];
                    else a0[q - NJ] = pA[SN][0][(q - NJ) * 128];
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            // ---- phase 1: k-half 0, phase 2: k-half 1 (accumulator n: strip n / 8 < NI, fragment n % 8 < NJ)
#define H_P1(n, R, C0, C1, C2, C3) \
            if ((n) < 8 * NI && ((n) & 7) < NJ) { W_MFMA(R, C0, C1, C2, C3, w0[(n) & 7], a0[((n) >> 3) & 3]); aux1(((n) >> 3) * NJ + ((n) & 7)); }
#define H_P2(n, R, C0, C1, C2, C3) \
            if ((n) < 8 * NI && ((n) & 7) < NJ) { W_MFMA(R, C0, C1, C2, C3, w1[(n) & 7], a1[((n) >> 3) & 3]); aux2(((n) >> 3) * NJ + ((n) & 7)); }
            W_FOR_EACH_ACC(H_P1)
            W_FOR_EACH_ACC(H_P2)
#undef H_P1
#undef H_P2
        };

        for (int kt = 0; kt < nk; kt += NST) {
            step(std::integral_constant<int, 0>{}, kt);
            if (kt + 1 >= nk) break;
            step(std::integral_constant<int, 1>{}, kt + 1);
            if (kt + 2 >= nk) break;
            step(std::integral_constant<int, 2>{}, kt + 2);
            if constexpr (NST == 4) {
                if (kt + 3 >= nk) break;
                step(std::integral_constant<int, 3>{}, kt + 3);
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> VALU reads (see the W_MFMA note)
#ifdef VR_W_TIMING
        tm2 = __builtin_amdgcn_s_memrealtime();
        cy2 = __builtin_amdgcn_s_memtime();
#endif
    }

    // ---- epilogue: 4 pieces of 32 rows x 16 NF columns per wave, through buffer descriptors (gemm256w_kernel.h has the why)
    auto uni32 = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    auto uni_ptr = [&](const void* q) {
        const unsigned long long a = (unsigned long long)q;
        return (void*)(((unsigned long long)uni32((unsigned)(a >> 32)) << 32) | uni32((unsigned)a));
    };
    constexpr int MI = 2;
    const int nb0 = n0 + wn * (16 * NJ);
    if constexpr (EPI == EPI_RESID) {
        const float alpha = __builtin_bit_cast(float, uni32(__builtin_bit_cast(unsigned, p.alpha)));
        const unsigned ldb = uni32((unsigned)p.ldo * 4u);
        const unsigned nrec = uni32((unsigned)min(H_BM, p.M - m0) * ldb);
        const auto ors = __builtin_amdgcn_make_buffer_rsrc(uni_ptr((float*)p.out + (size_t)m0 * p.ldo), 0, nrec, 0x00020000);
        const auto rrs = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(p.resid + (size_t)m0 * p.ldo), 0, nrec, 0x00020000);
        const auto brs = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(p.bias), 0, uni32(p.bias ? (unsigned)p.N * 4u : 0u), 0x00020000);
        const unsigned voff = (unsigned)(wm * 16 * NI + fr) * ldb + (unsigned)(nb0 + fq * 4) * 4u;
        f32x4 bias[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            bias[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (unsigned)(nb0 + fq * 4) * 4u + j * 64, 0, 0));
        // out = resid + alpha * (acc + bias), in place; the residual of piece q + RD is in flight while piece q is combined
        constexpr int RD = 2;
        f32x4 rs[RD + 1][MI][NF];
        auto load_piece = [&](int q, f32x4 (&dst)[MI][NF]) {
            const int h = q & 1, sg = q >> 1;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    dst[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        rrs, voff + (unsigned)((sg * MI + i) * 16) * ldb + (h * NF + j) * 64, 0, 0));
        };
#pragma unroll
        for (int q = 0; q < RD; ++q) load_piece(q, rs[q]);
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int h = q & 1, sg = q >> 1;
            if (q + RD < NI) load_piece(q + RD, rs[(q + RD) % (RD + 1)]);
            f32x4 acc[MI][NF];
#define H_RD(n, R, C0, C1, C2, C3) \
            if ((n) < 8 * NI && ((n) & 7) < NJ && ((n) & 7) / NF == h && ((n) >> 4) == sg) W_READ(acc[((n) >> 3) & 1][((n) & 7) % NF], C0, C1, C2, C3);
            W_FOR_EACH_ACC(H_RD)
#undef H_RD
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const f32x4 v = rs[q % (RD + 1)][i][j] + alpha * (acc[i][j] + bias[h * NF + j]);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors,
                                                           voff + (unsigned)((sg * MI + i) * 16) * ldb + (h * NF + j) * 64, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // lookup-free bf16 forms (NJ = 8): the tile's output rows behind one descriptor, the wave's bias columns loaded once
        static_assert(NF == 4, "64-column pieces");
        char* const wl0 = smem + NST * STAGE + wave * H_STAGING;
        const unsigned ldo2 = uni32((unsigned)p.ldo * 2u);
        const auto ors16 = __builtin_amdgcn_make_buffer_rsrc(uni_ptr((const char*)p.out + (size_t)m0 * p.ldo * 2), 0,
                                                             uni32((unsigned)min(H_BM, p.M - m0) * ldo2), 0x00020000);
        const auto brs = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(p.bias), 0, uni32(p.bias ? (unsigned)p.N * 4u : 0u), 0x00020000);
        const auto nors = __builtin_amdgcn_make_buffer_rsrc(nullptr, 0, 0u, 0x00020000);      // (no row bias here)
        f32x4 biasw[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            biasw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (unsigned)(nb0 + fq * 4) * 4u + j * 64, 0, 0));
        const f32x4 norope[MI][4] = {};
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int h = q & 1, sg = q >> 1;
            f32x4 acc[MI][NF];
#define H_RD(n, R, C0, C1, C2, C3) \
            if ((n) < 8 * NI && ((n) & 7) < NJ && ((n) & 7) / NF == h && ((n) >> 4) == sg) W_READ(acc[((n) >> 3) & 1][((n) & 7) % NF], C0, C1, C2, C3);
            W_FOR_EACH_ACC(H_RD)
#undef H_RD
            f32x4 b4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b4[j] = biasw[h * NF + j];
            gemm_epilogue_tile_lds_plain_buf<EPI, MI>(acc, b4, p, ors16, (unsigned)(wm * 16 * NI + sg * 32) * ldo2, ldo2, norope, nb0 + h * (16 * NF), lane, wl0,
                                                      nors, 0u);
            if (q & 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
#ifdef VR_W_TIMING
    if (threadIdx.x == 0 && p.rope_table) {
        const unsigned long long tm3 = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tm4 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* d = (unsigned long long*)p.rope_table + (size_t)blockIdx.x * 16;
        d[0] = tm0; d[1] = tm1; d[2] = tm2; d[3] = tm3; d[4] = tm4;
        d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID
        d[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // XCC_ID
        d[7] = cy2 - cy1;                                         // shader clocks spent in the K-loop
    }
#endif
}

// the LDS-DMA addresses whole matrices through 32-bit offsets, the epilogues a tile's rows likewise (gemm256w_fits)
bool gemm128w_fits(const GemmArgs& a, int bn) {
    const size_t tm = (a.M + H_BM - 1) / H_BM;
    return a.N > 0 && a.N % bn == 0 && a.K > 0 && a.K % GEMM_BK == 0 && !a.rowmap && !a.rowbias && a.ksplit <= 1 && !a.m_dev &&
           tm * H_BM * (size_t)a.lda * 2 < (1ull << 31) && (size_t)a.N * (size_t)a.ldw * 2 < (1ull << 31) &&
           (size_t)H_BM * (size_t)a.ldo * 4 < (1ull << 31);
}

template <int EPI, int NJ>
static hipError_t launch_h(GemmArgs a, hipStream_t s) {
    constexpr int BN = 32 * NJ;
    if (!gemm128w_fits(a, BN)) return hipErrorInvalidValue;
    if (EPI != EPI_RESID && ((a.ldo & 7) != 0 || a.col_scale_n != 0)) return hipErrorInvalidValue;
    const int tn = a.N / BN, tm = (a.M + H_BM - 1) / H_BM;
    // m-tiles per raster group: an XCD's share of the grid (total / 8 consecutive tiles) should be a squarish block — with
    // 17 x 12 tiles n-major order hands every XCD ALL rows of A (250 MB through the fabric for the down projection), groups
    // of 4 rows x all columns 150 MB
    if (a.raster_gm <= 0) a.raster_gm = tm <= 4 ? tm : 4;
#ifdef VR_W_TIMING
    if (const char* e = getenv("VR_H_GM")) a.raster_gm = atoi(e);
#endif
    constexpr int smem = h_smem_bytes<EPI, NJ>();
    static unsigned long long attr = 0;     // bit d: set on device d
    auto k = gemm128w_bf16_kernel<EPI, NJ>;
    set_max_dynamic_lds((const void*)k, smem, attr);
    hipLaunchKernelGGL(k, dim3(tn * tm), dim3(256), smem, s, a);
    return hipGetLastError();
}

// tile_cols 192: EPI_RESID; 256: EPI_RESID / EPI_BF16 / EPI_GELU / EPI_SWIGLU.  A rows readable up to the next multiple of
// 128, N % tile_cols == 0, no row map / row bias / split K / column scale.
hipError_t launch_gemm128w(const GemmArgs& a, int epi, int tile_cols, hipStream_t s) {
    if (tile_cols == 192) return epi == EPI_RESID ? launch_h<EPI_RESID, 6>(a, s) : hipErrorInvalidValue;
    if (tile_cols != 256) return hipErrorInvalidValue;
    switch (epi) {
        case EPI_RESID: return launch_h<EPI_RESID, 8>(a, s);
        case EPI_BF16: return launch_h<EPI_BF16, 8>(a, s);
        case EPI_GELU: return launch_h<EPI_GELU, 8>(a, s);
        case EPI_SWIGLU: return launch_h<EPI_SWIGLU, 8>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vr
