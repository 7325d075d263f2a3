// The band pass: what happens to a query whose candidate lists could not PROVE its fp32 top-k
// (search_common.h: certify_tail puts it on the flag list with its tau = s_k - eps and its bf16 row).
//
// Such a query needs every index row whose bf16-MFMA score is >= tau re-scored in fp32 — nothing else can belong
// to the fp32 top-k — and the fused sweeps cannot deliver that set when it is large (a cluster of near-duplicate
// pages: templated decks, forms) or when their lists dropped rows before tau was known (a pre-pass threshold
// inside the error band, a compacted half-list).  Rounds 1-3 sent these queries through an exact fp32 pass over the
// WHOLE index, one sweep of the fp32 rows per 8 queries (~1 ms per 8 queries at 100k rows).  Now:
//
//   1. S[slot][row] = bf16-MFMA score of flagged query `slot` against every row: ONE bf16 GEMM over the compacted
//      flagged queries (the 8-wave 256^2 kernel with a device-side row count: tiles past the flag count leave at
//      once, so the launch costs a dispatch when nothing is flagged) — the error bound holds for ANY fp32
//      accumulation order of the bf16 products, so these scores need not equal the sweep's bit for bit;
//   2. band_select_kernel (here), one workgroup (8 waves) per flagged query: ONE pass over its score row (16 bytes per
//      thread, four loads in flight) gathers the rows with S >= tau; each is re-scored in fp32 (dot_lane's chain: the one
//      definition of the library's fp32 score; four rows in flight per wave); the keys are reduced to the top k by
//      wave-level sorting networks (an LDS bitonic sort for k > 64).  Cost ~ band size x dim x 4 B of row reads.
//      (Measured and not kept, round 4: groups of four neighbouring flagged queries per workgroup — ordered by the row
//      id of their best candidate, the union of their bands loaded once, scored per query — were SLOWER on the templated
//      corpus, 2.0 against 1.7 ms for 993 flagged queries: the selection is a chain of latencies per workgroup — the
//      score-row pass, the row loads, the reductions — not the 9 GB of row reads, and a group serialises four chains.)
//   3. a band of more than BAND_MAX rows (8192: half of a 100k-row index scoring the same to 3e-3 is not a
//      retrieval problem any more) goes on a second list, and the exact fp32 pass (search_exact.hip) redoes it.
//
// Roofline: the GEMM is MFMA-bound (2 x flagged x rows x dim), the selection HBM / L2 (score rows + union rows).
#include "search_band.h"

namespace vr {

constexpr int BAND_NT = 512;        // threads (8 waves)

int search_band_max() { return BAND_MAX; }

__global__ __launch_bounds__(BAND_NT) void band_select_kernel(SearchArgs p, const float* __restrict__ S, size_t ldS, int sub,
                                                              int max_slots) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    BandLds& L = *reinterpret_cast<BandLds*>(smem_raw);
    __shared__ int ucnt_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int n_slots = min(max(p.flag_count[0] - sub, 0), max_slots);
    const int n_docs = (int)p.n_docs, k = p.k, dim = p.dim, nv = dim >> 2;
#ifndef BAND_XCD
#define BAND_XCD 0
#endif
    // (BAND_XCD: consecutive slots of a round on ONE XCD — workgroup b of the 256 takes slot (b % 8) * 32 + b / 8 of its round)
    const int bslot = BAND_XCD && gridDim.x == 256 ? (int)(blockIdx.x & 7) * 32 + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    for (int slot = bslot; slot < n_slots; slot += gridDim.x) {
        const int q = p.flag_list[sub + slot];
        const float tau = p.flag_tau[sub + slot];
        const float* row = S + (size_t)slot * ldS;
        // ---- 1. ONE pass over the score row: the band's row ids
        const int n_band = band_gather<BAND_NT>(row, 0, n_docs, tau, L, &ucnt_s);
        if (n_band > BAND_MAX || !(tau > -INFINITY)) {                   // workgroup-uniform: the exact fp32 pass ...
            if (p.exact_follows || k > 64) {
                if (tid == 0) {
                    const int pos = atomicAdd(p.flag2_count, 1);
                    p.flag2_list[pos] = q;
                    if (p.stats) atomicAdd(&p.stats[5], 1u);
                }
                continue;
            }
            // ... which is only launched behind an index that has shown such a band before (SearchArgs::huge_seen: its two idle
            // launches were 7 us of every search).  The first one is walked HERE, by this workgroup, in segments of BAND_MAX
            // rows (search_band.h: milliseconds per query), and sets the word.
            __syncthreads();
            band_pass_in_place<BAND_NT>(p, q, tau, row, L, &ucnt_s);
            continue;
        }
        // ---- 2. exact fp32 scores
        f32x4 qv[MERGE_MAXV];
        load_query_regs(qv, p.q_f32 + (size_t)q * dim, nv, lane);
        band_rescore<BAND_NT>(p, qv, n_band, L);
        // ---- 3. the top k
        uint64_t* kq = L.keys;
        if (k <= 64) {
            // the best 64 without a workgroup barrier per sorting stage: every wave folds its share of the keys into a sorted
            // top-64 (register network), wave 0 merges the eight lists
            uint64_t best = KEY_NONE;
            bool first = true;
            band_fold64<BAND_NT>(L, n_band, best, first);
            band_emit64<BAND_NT>(p, q, L, best, first);
            continue;
        }
        int n2 = 64;
        while (n2 < n_band) n2 <<= 1;
        for (int e = n_band + tid; e < n2; e += BAND_NT) kq[e] = KEY_NONE;
        __syncthreads();
        block_bitonic_desc(kq, n2, tid, BAND_NT);
        for (int e = tid; e < k; e += BAND_NT) emit_slot(p, q, e, e < n_band ? kq[e] : KEY_NONE);
    }
}

hipError_t launch_band_select(const SearchArgs& a, const float* S, size_t ldS, int sub, int max_slots, hipStream_t s) {
    if (max_slots <= 0 || a.n_docs <= 0) return hipSuccess;
    if (a.dim % 4 || a.dim > 64 * 4 * MERGE_MAXV || a.k > BAND_MAX || !a.flag_count || !a.flag_list || !a.flag_tau || !a.flag2_count ||
        !a.flag2_list)
        return hipErrorInvalidValue;
    static unsigned long long attr = 0;     // bit d: set on device d
    set_max_dynamic_lds((const void*)band_select_kernel, (int)sizeof(BandLds), attr);
    // a fixed grid walks the flagged slots (one workgroup per CU: 96 KiB of LDS); all leave at once when nothing is flagged
    hipLaunchKernelGGL(band_select_kernel, dim3(max_slots < 256 ? max_slots : 256), dim3(BAND_NT), sizeof(BandLds), s, a, S, ldS, sub,
                       max_slots);
    return hipGetLastError();
}

}  // namespace vr
