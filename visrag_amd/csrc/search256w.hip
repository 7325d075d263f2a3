// Search sweep on the one-wave-per-SIMD 256 x 256 x 64 tile (see gemm256w.hip): the main sweep of
// vr_index_search for more than 128 queries when dim is a multiple of 128.
//
// What it computes and how candidates are kept is search256.hip's (fused similarity GEMM, scores
// filtered in registers against a per-query threshold, survivors appended to wave-owned half-lists
// in a global scratch, lengths replicated in registers, no atomics; the same merge kernel reads the
// lists).  What changes is the main loop and the shape of a wave's share:
//
//   workgroup = (doc chunk, 256-query tile), 4 waves (2 x 2); A = index rows (docs), W = query rows.
//   wave (wm, wn): docs wm*128 + i*16 + fr (8 strips), queries wn*128 + j*16 + fq*4 + r (8 fragments);
//   its 256 accumulators live in hand-allocated accumulation registers (gemm256w_acc.h).
//
// ONE K-step stream over all tiles of the chunk, loads two K-steps ahead across tile boundaries: the
// two steps "past the end" of a tile are the first two of the next (last tile: out of the descriptor's
// range), the last step's second phase already reads the next tile's first fragments, so after the
// filter epilogue the MFMA stream resumes at once.  Index rows come from HBM (each is read once per
// chunk and shared by the query tiles of the chunk through the XCD's L2); search256.hip's touch-ahead of
// step s + 3 measured -7 % here (a wave's loads complete in order, so the touch has one K-step to land
// and the K-step is now 1.1 us).  Per tile (1000 queries x 100k rows): K-loop 44-48 us, filter epilogue 12 us with the
// 4096-row pre-pass and a compare -> ballot -> branch per value, 8 us with the sign-mask form below (round 6).
// Round 6 also: when the threshold pre-pass OWNS its sampled tiles (SearchArgs::pre_own_chunks, search.hip) this sweep
// walks the OTHER index tiles only — 375 instead of 391 at 100k rows: six per workgroup instead of seven.
//
// Half-list of query qq, owner half wm: [query][chunk][wm][64] keys, as in search256.hip; a wave now
// owns the half-lists of 128 queries (32 byte-counters per lane in 8 registers).
// Roofline: MFMA (2 * Nq * Nd * dim flop per sweep).
#include <cstdlib>
#include <type_traits>

#include "gemm_core.h"
#include "gemm256w_acc.h"
#include "kernels.h"
#include "search_common.h"

namespace vr {

namespace {

constexpr int SW_STAGE = 2 * G256_TILE_BYTES;      // A tile + W tile = 64 KiB
constexpr unsigned SW_OOB = 0x80000000u;
constexpr int SW_HL_CAP = 64;                      // slots per half-list (= search256.hip's)
constexpr int SW_HL_TRIG = 48;                     // compact a half-list longer than this (a strip adds <= 16)
constexpr int SW_SMEM = 2 * SW_STAGE + 256 * 4 + 256;    // stages + thresholds + per-query "a list was compacted" bytes
// timing diagnostics only (tagged builds, tools/variant.sh; results are wrong): 1 no filter epilogue, 2 filter without its stores / counters
#ifndef SW_DBG
#define SW_DBG 0
#endif

__device__ __forceinline__ uint64_t sw_ld_key(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sw_st_key(unsigned long long* p, uint64_t v) {
    __hip_atomic_store(p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// OR of a 32-bit value over the wave, as a scalar: row_shr 1 / 2 / 4 / 8 leave a row's OR in its lane 15, row_bcast 15 / 31
// carry it on to lane 63 (OR is idempotent: no bank masks needed)
__device__ __forceinline__ uint32_t sw_wave_or(uint32_t x) {
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);     // row_shr:1
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);     // row_shr:2
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);     // row_shr:4
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);     // row_shr:8
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

__device__ __forceinline__ void* sw_uniform_ptr(const char* q) {
    const uint64_t v = (uint64_t)q;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (void*)(((uint64_t)hi << 32) | lo);
}

}  // namespace

template <int KP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void search_sweep256w_kernel(SearchArgs p, int q_tiles, int tiles_per_chunk, const float* __restrict__ thr_init, int pre_spots, int pre_step) {
    constexpr int NS = 64, NR = 16, DS = 5, SB1 = 40, D1 = 5, SB2 = 8;     // the schedule of gemm256w.hip, NJ = 8
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const thr_lds = reinterpret_cast<float*>(smem + 2 * SW_STAGE);
    unsigned char* const comp_lds = reinterpret_cast<unsigned char*>(smem + 2 * SW_STAGE + 256 * 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;

    const int b = blockIdx.x;
    const int chunk = __builtin_amdgcn_readfirstlane((b / (8 * q_tiles)) * 8 + (b & 7));   // the query tiles of a chunk share an XCD
    const int qt = __builtin_amdgcn_readfirstlane((b >> 3) % q_tiles);
    const int q0 = qt * 256;
    // the tiles this sweep walks, numbered 0 .. n_tiles - 1: all index tiles, or (pre_spots > 0) all but the pre-pass's sampled
    // ones — index tile k * pre_step + pre_step - 1 for k < pre_spots is scored by search_prepass_own_kernel and nobody else
    const int n_tiles = (int)((p.n_docs + 255) / 256) - pre_spots;
    const int tile_lo = chunk * tiles_per_chunk;
    const int tile_hi = min(n_tiles, tile_lo + tiles_per_chunk);
    auto index_tile = [&](int t) { return pre_spots > 0 ? t + min(t / (pre_step - 1), pre_spots) : t; };
    // half-list of query qq (0..255 in this tile), owner half wm: gw + qq * gq
    unsigned long long* gw = p.cand_keys + (((size_t)q0 * p.n_chunks + chunk) * 2 + wm) * SW_HL_CAP;
    const size_t gq = (size_t)p.n_chunks * 2 * SW_HL_CAP;

    {
        float t0 = (thr_init && q0 + tid < p.nq) ? thr_init[q0 + tid] : -INFINITY;
        if (q0 + tid >= p.nq) t0 = INFINITY;                  // padding queries never collect candidates
        thr_lds[tid] = t0;
        comp_lds[tid] = 0;
    }
    __syncthreads();

    // lengths of this wave's half-lists, replicated over the 16 lanes of a group:
    // byte r of c8[j] = length of the list of query wn*128 + j*16 + fq*4 + r
    // (a register VECTOR, not an array: indexed with a runtime j in the rare compaction path below, an array
    // ends up in scratch, and every scratch access of the epilogue then waits for the loads in flight)
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    u32x8 c8 = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    auto c8_get = [&](int j) {
        uint32_t v = c8[0];
#pragma unroll
        for (int t = 1; t < 8; ++t) v = j == t ? c8[t] : v;
        return v;
    };
    auto c8_set = [&](int j, uint32_t v) {
#pragma unroll
        for (int t = 0; t < 8; ++t) c8[t] = j == t ? v : c8[t];
    };

    // rare: a half-list of this wave passed SW_HL_TRIG -> the wave alone sorts it, keeps the best KP and raises
    // the query's threshold (a runtime loop over the 32 (j, r) columns, kept rolled)
    auto compact_own = [&]() {
        __threadfence_block();                                  // own stores visible to own loads
#pragma nounroll
        for (int jr = 0; jr < 32; ++jr) {
            const int j = jr >> 2, r = jr & 3;
            const uint32_t cw = c8_get(j);
            const int c_l = (int)((cw >> (8 * r)) & 0xFFu);
            unsigned long long todo = __ballot(c_l > SW_HL_TRIG && fr == 0);   // one bit per group
            uint32_t nw = cw;
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;                // lane fq*16
                todo &= todo - 1;
                const int qq = wn * 128 + j * 16 + (src >> 4) * 4 + r;
                const int c = __shfl(c_l, src, 64);
                unsigned long long* row = gw + (size_t)qq * gq;
                uint64_t key = (lane < c) ? sw_ld_key(row + lane) : KEY_NONE;
                key = wave_sort_desc(key);
                if (lane < KP) sw_st_key(row + lane, key);
                const int keep = min(c, KP);
                if (lane == KP - 1 && c >= KP) {
                    const float t = orderable_f32((uint32_t)(key >> 32));
                    if (t > thr_lds[qq]) thr_lds[qq] = t;          // (the partner half may race: both bounds are valid)
                }
                if (fq == (src >> 4)) nw = (nw & ~(0xFFu << (8 * r))) | ((uint32_t)keep << (8 * r));
                if (lane == 0) comp_lds[qq] = 1;                   // rows were dropped from this query's lists: the merge must know
            }
            c8_set(j, nw);
        }
        __threadfence_block();
    };

    if (tile_lo < tile_hi) {
        // ---- LDS-DMA addressing (gemm256w.hip): descriptors at this workgroup's first index tile / its query
        //      tile; A and W share the row pitch.  The launcher keeps a chunk below 2 GiB.
        const int nk = p.dim / GEMM_BK;
        const unsigned tile_bytes = 256u * (unsigned)p.dim * 2u;
        // (pinned into SGPRs: a descriptor hipcc cannot prove wave-uniform gets a readfirstlane loop around every load)
        const auto arsrc = __builtin_amdgcn_make_buffer_rsrc(
            sw_uniform_ptr((const char*)p.index_bf16 + (size_t)index_tile(tile_lo) * tile_bytes), 0, 0x7FFFFFFF, 0x00020000);
        const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc(
            sw_uniform_ptr((const char*)p.q_bf16 + (size_t)q0 * p.dim * 2), 0, 0x7FFFFFFF, 0x00020000);
        const unsigned lof = (unsigned)(lane >> 3) * (unsigned)p.dim * 2u + (unsigned)(((lane & 7) ^ (lane >> 3)) << 4);
        const unsigned rg = (unsigned)p.dim * 16u;            // bytes per 8-row group
        const unsigned s0 = (unsigned)wave * 8u * rg;
        char* const dmaA = smem + wave * 8192;
        char* const dmaW = smem + G256_TILE_BYTES + wave * 8192;
        auto dma = [&](int stage, int d, unsigned vA, unsigned vW) {
            if (d < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, VR_LDS(dmaA + stage * SW_STAGE + d * 1024), 16, vA, s0 + d * rg, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, VR_LDS(dmaW + stage * SW_STAGE + (d - 8) * 1024), 16, vW, s0 + (d - 8) * rg, 0, 0);
        };
        typedef const __attribute__((address_space(3))) bf16x8* frag_p;
        frag_p pA[2][2], pW[2][2];
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ch = ((kk * 4 + fq) ^ (fr & 7)) << 4;
                pA[st][kk] = (frag_p)VR_LDS(smem + st * SW_STAGE + (wm * 128 + fr) * 128 + ch);
                pW[st][kk] = (frag_p)VR_LDS(smem + st * SW_STAGE + G256_TILE_BYTES + (wn * 128 + fr) * 128 + ch);
                asm volatile("" : "+v"(pA[st][kk]), "+v"(pW[st][kk]));
            }

        bf16x8 a0[8], w0[8], a1[8], w1[8];
        {   // prologue: K-steps 0 and 1 of the first tile in flight, k-half-0 fragments of step 0 requested
#pragma unroll
            for (int d = 0; d < NR; ++d) dma(0, d, lof, lof);
#pragma unroll
            for (int d = 0; d < NR; ++d) dma(1, d, lof + (unsigned)(GEMM_BK * 2), lof + (unsigned)(GEMM_BK * 2));
            W_FOR_EACH_ACC(W_ZERO)                              // (under the loads' latency)
            VR_WAIT_VM_BARRIER(16);
#pragma unroll
            for (int j = 0; j < 8; ++j) w0[j] = pW[0][0][j * 128];
#pragma unroll
            for (int i = 0; i < 8; ++i) a0[i] = pA[0][0][i * 128];
        }

        for (int tile = tile_lo; tile < tile_hi; ++tile) {
            const int itile = index_tile(tile);
            const unsigned curA = (unsigned)(itile - index_tile(tile_lo)) * tile_bytes;
            const bool more = tile + 1 < tile_hi;
            const unsigned nextA = (unsigned)(index_tile(tile + 1) - index_tile(tile_lo)) * tile_bytes;    // (past a skipped tile)
            auto step = [&](auto stage_c, int kt) {
                constexpr int S = decltype(stage_c)::value;
                const int k2 = kt + 2;
                // step kt + 2 of this tile, or step kt + 2 - nk of the next one, or (last tile) nowhere
                const unsigned kb = (unsigned)(k2 < nk ? k2 : k2 - nk) * (GEMM_BK * 2);
                const unsigned vA = lof + (k2 < nk ? curA + kb : (more ? nextA + kb : SW_OOB));
                const unsigned vW = lof + ((k2 < nk || more) ? kb : SW_OOB);
                __builtin_amdgcn_sched_barrier(0);
                auto aux1 = [&](int sl) {
                    if (sl < 2 * NR && (sl & 1) == 0) {
                        const int q = sl >> 1;
                        if (q < 8) w1[q] = pW[S][1][q * 128];
                        else a1[q - 8] = pA[S][1][(q - 8) * 128];
                    }
                    if (sl == SB1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    if (sl >= SB1 + 2 && (sl - (SB1 + 2)) % DS == 0) dma(S, (sl - (SB1 + 2)) / DS, vA, vW);
                    __builtin_amdgcn_sched_barrier(0);
                };
                auto aux2 = [&](int sl) {
                    if (sl == SB2) VR_WAIT_VM_BARRIER(5);
                    if (sl >= SB2 + 3 && (sl - (SB2 + 3)) % DS == 0 && D1 + (sl - (SB2 + 3)) / DS < NR)
                        dma(S, D1 + (sl - (SB2 + 3)) / DS, vA, vW);
                    if (sl >= SB2 + 2 && sl < SB2 + 2 + 2 * NR && ((sl - SB2) & 1) == 0) {
                        const int q = (sl - (SB2 + 2)) >> 1;
                        if (q < 8) w0[q] = pW[S ^ 1][0][q * 128];
                        else a0[q - 8] = pA[S ^ 1][0][(q - 8) * 128];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                static_assert(SB2 + 3 + DS * (NR - D1 - 1) < NS && SB2 + 2 + 2 * (NR - 1) < NS && 2 * NR - 2 < SB1, "schedule fits");
#define SW_P1(n, R, C0, C1, C2, C3) W_MFMA(R, C0, C1, C2, C3, w0[(n) & 7], a0[(n) >> 3]); aux1(n);
#define SW_P2(n, R, C0, C1, C2, C3) W_MFMA(R, C0, C1, C2, C3, w1[(n) & 7], a1[(n) >> 3]); aux2(n);
                W_FOR_EACH_ACC(SW_P1)
                W_FOR_EACH_ACC(SW_P2)
#undef SW_P1
#undef SW_P2
            };
            for (int kt = 0; kt < nk; kt += 2) {            // (nk is even: the launcher checks dim % 128 == 0)
                step(std::integral_constant<int, 0>{}, kt);
                step(std::integral_constant<int, 1>{}, kt + 1);
            }
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> VALU reads

            // ---- filter epilogue of this tile, strip by strip (32 accumulator registers read back at a time; no
            //      workgroup barrier).  acc n = strip * 8 + fragment: v[j][r] = score(doc strip i, query j*16 + fq*4 + r)
            const int doc0 = itile * 256;
            const uint32_t below = (1u << fr) - 1u;
#pragma unroll
            for (int i = 0; i < ((SW_DBG & 1) ? 0 : 8); ++i) {
                f32x4 v[8];
#define SW_RD(n, R, C0, C1, C2, C3) if (((n) >> 3) == i) W_READ(v[(n) & 7], C0, C1, C2, C3);
                W_FOR_EACH_ACC(SW_RD)
#undef SW_RD
                const int doc = doc0 + wm * 128 + i * 16 + fr;
                const bool valid = doc < p.n_docs;
                // The lane's 32 scores of the strip against their thresholds WITHOUT a branch or a scalar-register round trip per
                // value: the sign of s - thr is shifted into a mask (v_sub + v_alignbit, bit 31 - (j * 4 + r) = "below"), the
                // wave's OR of the pass masks comes back as ONE scalar (six DPP steps + a readlane), and only the (j, r) columns
                // whose bit is set in it — ~9 of 32 with the 4096-row pre-pass — take the ballot / append path.  Round 6,
                // tools/r6/sweep_anatomy.sh: a compare -> ballot -> `s_cbranch_vccz` per value (v_cmp, s_and, v_cndmask, v_cmp_ne,
                // branch: every step waits for the other pipe) cost 6.4 us of a tile's 12.2 us filter, the appends 5.3.
                uint32_t mlow = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const f32x4 th = *reinterpret_cast<const f32x4*>(&thr_lds[wn * 128 + j * 16 + fq * 4]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) mlow = __builtin_amdgcn_alignbit(mlow, __builtin_bit_cast(uint32_t, v[j][r] - th[r]), 31);
                }
                const uint32_t mpass = valid ? ~mlow : 0u;
                const uint32_t um = (SW_DBG & 2) ? 0u : sw_wave_or(mpass);
                if (um) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int qn = wn * 128 + j * 16 + fq * 4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const uint32_t bit = 0x80000000u >> (j * 4 + r);
                            if (um & bit) {                         // scalar test: some lane of the wave has a survivor in column (j, r)
                                const bool pass = (mpass & bit) != 0u;
                                const unsigned long long bal = __ballot(pass);
                                const uint32_t grp = (uint32_t)(bal >> (fq * 16)) & 0xFFFFu;
                                if (pass) {
                                    const int pos = (int)((c8[j] >> (8 * r)) & 0xFFu) + __popc(grp & below);
                                    sw_st_key(gw + (size_t)(qn + r) * gq + pos, make_key(v[j][r], (uint32_t)doc));
                                }
                                c8[j] += (uint32_t)__popc(grp) << (8 * r);
                            }
                        }
                    }
                }
                bool over = false;
#pragma unroll
                for (int j = 0; j < 8; ++j) over |= ((c8[j] + 0x01010101u * (127 - SW_HL_TRIG)) & 0x80808080u) != 0u;
                if (__any(over)) compact_own();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) { W_FOR_EACH_ACC(W_ZERO) }
        }
    }
    // ---- list lengths: [query][chunk][wm]
    if (fr == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = wn * 128 + j * 16 + fq * 4 + r;
                p.cand_ids[((size_t)(q0 + qq) * p.n_chunks + chunk) * 2 + wm] =
                    (int)((c8[j] >> (8 * r)) & 0xFFu) | (comp_lds[qq] ? 0x100 : 0);   // 0x100 = search256.hip's HL_COMPACTED
            }
        }
    }
}

// sweep geometry: workgroup chunks, the tiles they walk (index tiles minus the pre-pass's own), tiles per chunk
namespace {
struct SwGeo { int chunks, n_tiles, spots, step, tpc; size_t span_bytes; };
SwGeo sw_geometry(const SearchArgs& a) {
    SwGeo g{};
    const int all = (int)((a.n_docs + 255) / 256);
    g.spots = a.pre_own_chunks > 0 ? SEARCH_PRE_SPOTS : 0;
    g.step = g.spots ? all / g.spots : 0;
    g.n_tiles = all - g.spots;
    g.chunks = a.n_chunks - a.pre_own_chunks;
    g.tpc = g.chunks > 0 ? (g.n_tiles + g.chunks - 1) / g.chunks : 0;
    // index tiles a chunk's 32-bit offsets may reach: its own, the next one (prefetch), the skipped ones in between
    const int skipped = g.spots ? g.tpc / (g.step - 1) + 2 : 0;
    g.span_bytes = (size_t)(g.tpc + 1 + skipped) * 256 * a.dim * 2;
    return g;
}
}  // namespace

template <int KP>
static hipError_t launch_t(const SearchArgs& a, const float* thr, hipStream_t s) {
    const int q_tiles = (a.nq + 255) / 256;
    const SwGeo g = sw_geometry(a);
    if (g.chunks <= 0 || g.chunks % 8 || g.span_bytes >= (1ull << 31)) return hipErrorInvalidValue;      // 32-bit chunk offsets
    if (g.spots && (g.step < 8 || !a.thr_cert)) return hipErrorInvalidValue;
    auto k = search_sweep256w_kernel<KP>;
    static unsigned long long attr = 0;     // bit d: set on device d
    set_max_dynamic_lds((const void*)k, SW_SMEM, attr);
    hipLaunchKernelGGL(k, dim3(g.chunks * q_tiles), dim3(256), SW_SMEM, s, a, q_tiles, g.tpc, thr, g.spots, g.step);
    return hipGetLastError();
}

// the sweep only (same scratch layout as launch_sweep256, whose merge kernel follows): dim % 128 == 0
bool sweep256w_ok(const SearchArgs& a) {
    const SwGeo g = sw_geometry(a);
    return a.dim % 128 == 0 && g.chunks > 0 && g.span_bytes < (1ull << 31);
}
hipError_t launch_sweep256w(const SearchArgs& a, int kp, const float* thr, hipStream_t s) {
    if (!a.cand_keys || a.n_chunks % 8 || !sweep256w_ok(a)) return hipErrorInvalidValue;
    switch (kp) {
        case 16: return launch_t<16>(a, thr, s);
        case 32: return launch_t<32>(a, thr, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vr
