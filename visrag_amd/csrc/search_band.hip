// The band pass: what happens to a query whose candidate lists could not PROVE its fp32 top-k
// (search_common.h: certify_tail puts it on the flag list with its tau = s_k - eps, its bf16 row and its best row id).
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
//   2. band_select_kernel (here).  The flagged queries are ordered by the row id of their best candidate — queries
//      that hit the same family of near-duplicates become neighbours — and a workgroup takes a GROUP of BAND_G of
//      them: one pass over their score rows gathers the UNION of their bands (row + a bit per query whose band holds
//      it), every union row is loaded ONCE and its fp32 score taken for each of those queries (dot_lane's chain, the
//      queries' fp32 rows in LDS: the one definition of the library's fp32 score), the keys of a query are sorted
//      in LDS and its top k emitted.  (First version: one query per workgroup — on a templated corpus, 993 flagged
//      queries x ~1 000 band rows x 9 KB = 9 GB of row reads, 1.5 ms at the memory's pace; queries of one family
//      share those rows.)  The queries of a group whose bands hold at most 2048 rows share a round; a larger band
//      (up to BAND_MAX rows) takes a round — the whole key buffer — of its own;
//   3. a band of more than BAND_MAX rows (8192: half of a 100k-row index scoring the same to 3e-3 is not a
//      retrieval problem any more) goes on a second list, and the exact fp32 pass (search_exact.hip) redoes it.
//
// Roofline: the GEMM is MFMA-bound (2 x flagged x rows x dim), the selection HBM / L2 (score rows + union rows).
#include "kernels.h"
#include "search_common.h"

namespace vr {

constexpr int BAND_MAX = 8192;      // rows a single query's band may hold (its keys: 64 KiB of LDS)
constexpr int BAND_G = 4;           // queries per group
constexpr int BAND_KQ = BAND_MAX / BAND_G;   // band rows per query when the whole group shares a round (2048)
constexpr int BAND_NT = 512;        // threads (8 waves, two rows in flight each)
constexpr int BAND_SORT = 4096;     // flagged queries ordered per pass (more: processed in slot order)

int search_band_max() { return BAND_MAX; }

struct BandLds {                    // dynamic LDS image: 144 KiB, one workgroup per CU
    uint64_t keys[BAND_MAX];                                     // 64 KiB: the round's keys [queries][stride] | the slot sort
    uint32_t cand[BAND_MAX];                                     // union rows of a round (<= the sum of its queries' bands <= BAND_MAX)
    uint8_t cmask[BAND_MAX];                                     // bit g: row belongs to the band of the group's query g
    float q[BAND_G * 64 * 4 * MERGE_MAXV];                       // the group's fp32 query rows (dim <= 2560)
};

__global__ __launch_bounds__(BAND_NT) void band_select_kernel(SearchArgs p, const float* __restrict__ S, size_t ldS, int sub,
                                                              int max_slots) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    BandLds& L = *reinterpret_cast<BandLds*>(smem_raw);
    __shared__ int ucnt_s, gcnt_s[BAND_G], gband_s[BAND_G], gslot_s[BAND_G], gq_s[BAND_G];
    __shared__ float gtau_s[BAND_G];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = BAND_NT / 64;
    const int n_slots = min(max(p.flag_count[0] - sub, 0), max_slots);
    if (n_slots <= 0) return;
    const int n_docs = (int)p.n_docs, k = p.k, dim = p.dim, nv = dim >> 2;
    // queries per group: BAND_G when there are enough flagged queries to fill the chip with groups, else one (few flagged
    // queries: a workgroup each — sharing rows would leave CUs idle)
    const int gsz = n_slots > 2 * 256 ? BAND_G : 1;
    const int n_groups = (n_slots + gsz - 1) / gsz;
    if ((int)blockIdx.x >= n_groups) return;

    // ---- slots ordered by the row id of the query's best candidate (every workgroup sorts the few thousand words itself:
    //      cheaper than a launch that would also be issued when nothing is flagged); beyond BAND_SORT flagged: slot order
    const bool sorted = n_slots <= BAND_SORT && gsz > 1;
    int n2s = 64;
    while (n2s < n_slots) n2s <<= 1;

    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        __syncthreads();                                                 // (LDS of the previous group is free)
        if (sorted) {
            for (int e = tid; e < n2s; e += BAND_NT)
                L.keys[e] = e < n_slots ? (((uint64_t)(~p.flag_top[sub + e]) << 32) | (uint32_t)e) : KEY_NONE;   // descending sort = ascending row id
            __syncthreads();
            block_bitonic_desc(L.keys, n2s, tid, BAND_NT);
        }
        const int ng = min(gsz, n_slots - grp * gsz);
        if (tid < BAND_G) {
            const int sl = tid < ng ? (sorted ? (int)(uint32_t)L.keys[grp * gsz + tid] : grp * gsz + tid) : -1;
            gslot_s[tid] = sl;
            gq_s[tid] = sl >= 0 ? p.flag_list[sub + sl] : -1;
            gtau_s[tid] = sl >= 0 ? p.flag_tau[sub + sl] : INFINITY;     // (an absent query: nothing passes)
            gband_s[tid] = 0;
        }
        __syncthreads();
        int slot[BAND_G];
        float tau[BAND_G];
#pragma unroll
        for (int g = 0; g < BAND_G; ++g) { slot[g] = gslot_s[g]; tau[g] = gtau_s[g]; }
        // ---- 0. size of every query's band; the queries' fp32 rows -> LDS
        {
            int c[BAND_G] = {0, 0, 0, 0};
#pragma unroll
            for (int g = 0; g < BAND_G; ++g) {
                if (slot[g] < 0) continue;                               // workgroup-uniform
                const float* row = S + (size_t)slot[g] * ldS;
                int i = tid;
                for (; i + 7 * BAND_NT < n_docs; i += 8 * BAND_NT) {     // eight independent loads in flight
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = row[i + u * BAND_NT];
#pragma unroll
                    for (int u = 0; u < 8; ++u) c[g] += v[u] >= tau[g] ? 1 : 0;
                }
                for (; i < n_docs; i += BAND_NT) c[g] += row[i] >= tau[g] ? 1 : 0;
            }
#pragma unroll
            for (int g = 0; g < BAND_G; ++g)
                if (c[g]) atomicAdd(&gband_s[g], c[g]);
            for (int e = tid; e < ng * nv; e += BAND_NT) {
                const int g = e / nv, cc = e % nv;
                reinterpret_cast<f32x4*>(L.q)[g * nv + cc] = reinterpret_cast<const f32x4*>(p.q_f32 + (size_t)gq_s[g] * dim)[cc];
            }
        }
        __syncthreads();
        // ---- rounds: first every query whose band fits a quarter of the key buffer, together; then the larger ones (up to
        //      BAND_MAX rows) one at a time; a band beyond that, or an unknown tau, goes to the exact pass's list
        unsigned small = 0u, todo = 0u;
#pragma unroll
        for (int g = 0; g < BAND_G; ++g) {
            if (slot[g] < 0) continue;
            const int nb = gband_s[g];
            if (!(tau[g] > -INFINITY) || nb > BAND_MAX) {
                if (tid == 0) {
                    const int pos = atomicAdd(p.flag2_count, 1);
                    p.flag2_list[pos] = gq_s[g];
                    if (p.stats) atomicAdd(&p.stats[5], 1u);
                }
            } else if (nb <= BAND_KQ) small |= 1u << g;
            else todo |= 1u << g;
        }
#pragma nounroll
        for (int round = 0; round <= BAND_G; ++round) {
            unsigned sel;                                                // the queries of this round (workgroup-uniform)
            if (round == 0) sel = small;
            else { sel = todo & (0u - todo); todo &= todo - 1u; }       // lowest remaining large one
            if (sel == 0u) { if (round > 0 && todo == 0u) break; continue; }
            const int stride = round == 0 ? BAND_KQ : BAND_MAX;
            __syncthreads();
            if (tid < BAND_G) gcnt_s[tid] = 0;
            if (tid == 0) ucnt_s = 0;
            __syncthreads();
            // ---- 1. the union of the round's bands: row ids + a bit per query
            for (int i0 = 0; i0 < n_docs; i0 += 4 * BAND_NT) {
                unsigned m4[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int g = 0; g < BAND_G; ++g) {
                    if (!((sel >> g) & 1u)) continue;                    // workgroup-uniform
                    const float* row = S + (size_t)slot[g] * ldS;
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int i = i0 + u * BAND_NT + tid; v[u] = i < n_docs ? row[i] : -INFINITY; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (v[u] >= tau[g]) m4[u] |= 1u << g;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned m = m4[u];
                    const unsigned long long b = __ballot(m != 0u);
                    if (b == 0ull) continue;                             // wave-uniform
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&ucnt_s, __popcll(b));
                    base = __shfl(base, 0, 64);
                    const int pos = base + __popcll(b & ((1ull << lane) - 1ull));
                    if (m != 0u && pos < BAND_MAX) { L.cand[pos] = (uint32_t)(i0 + u * BAND_NT + tid); L.cmask[pos] = (uint8_t)m; }
                }
            }
            __syncthreads();
            const int nu = min(ucnt_s, BAND_MAX);                        // (<= the sum of the round's bands <= BAND_MAX by construction)
            // ---- 2. every union row once: its fp32 score for each query whose band holds it
            const f32x4* qs = reinterpret_cast<const f32x4*>(L.q);
            auto load_row = [&](f32x4 (&dv)[MERGE_MAXV], int cidx) {
                if (cidx >= nu) return;
                const f32x4* dr = reinterpret_cast<const f32x4*>(p.index_f32 + (size_t)L.cand[cidx] * dim);
#pragma unroll
                for (int i = 0; i < MERGE_MAXV; ++i) {
                    const int cc = lane + i * 64;
                    dv[i] = (cc < nv) ? dr[cc] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            };
            auto score_row = [&](const f32x4 (&dv)[MERGE_MAXV], int cidx) {
                const unsigned m = L.cmask[cidx];
                const uint32_t id = L.cand[cidx];
                const f32x4* qp = qs;
                asm volatile("" : "+v"(qp));                             // (the queries are re-read from LDS per row: hoisted out of the
                                                                         // row loop they would take 160 registers)
#pragma unroll
                for (int g = 0; g < BAND_G; ++g) {
                    if (!((m >> g) & 1u)) continue;                      // wave-uniform
                    float a = 0.f;
#pragma unroll
                    for (int i = 0; i < MERGE_MAXV; ++i) {
                        const int cc = lane + i * 64;
                        if (cc < nv) a = dot_chunk(qp[g * nv + cc], dv[i], a);          // the chain of dot_lane(q, row)
                    }
                    a = wave_sum(a);
                    if (lane == 0) {
                        const int pos = atomicAdd(&gcnt_s[g], 1);
                        if (pos < stride) L.keys[(round == 0 ? g * BAND_KQ : 0) + pos] = make_key(a, id);
                    }
                }
            };
            f32x4 dA[MERGE_MAXV], dB[MERGE_MAXV];
            load_row(dA, wave);
#pragma nounroll
            for (int cidx = wave; cidx < nu; cidx += 2 * NW) {
                load_row(dB, cidx + NW);
                score_row(dA, cidx);
                if (cidx + NW >= nu) break;
                load_row(dA, cidx + 2 * NW);
                score_row(dB, cidx + NW);
            }
            __syncthreads();
            // ---- 3. per query of the round: sort its keys, emit
            for (int g = 0; g < BAND_G; ++g) {
                if (!((sel >> g) & 1u)) continue;
                const int n = min(gcnt_s[g], stride);
                uint64_t* kq = L.keys + (round == 0 ? g * BAND_KQ : 0);
                if (k <= 64) {
                    // the best 64 without a workgroup barrier per sorting stage: every wave folds its share of the keys into a
                    // sorted top-64 (register network), wave 0 merges the eight lists
                    uint64_t best = KEY_NONE;
                    for (int base = wave * 64; base < n; base += NW * 64) {
                        const uint64_t key = base + lane < n ? kq[base + lane] : KEY_NONE;
                        best = base == wave * 64 ? wave_sort_desc(key) : wave_merge_top64(best, key, lane);
                    }
                    __syncthreads();                                     // (everyone has read its keys: the head of kq is reused)
                    kq[wave * 64 + lane] = best;
                    __syncthreads();
                    if (wave == 0) {
                        uint64_t top = kq[lane];
                        for (int w = 1; w < NW; ++w) top = wave_merge_top64(top, kq[w * 64 + lane], lane);
                        if (lane < k) emit_slot(p, gq_s[g], lane, top);
                    }
                    __syncthreads();
                    continue;
                }
                int n2 = 64;
                while (n2 < n) n2 <<= 1;
                for (int e = n + tid; e < n2; e += BAND_NT) kq[e] = KEY_NONE;
                __syncthreads();
                block_bitonic_desc(kq, n2, tid, BAND_NT);
                for (int e = tid; e < k; e += BAND_NT) emit_slot(p, gq_s[g], e, e < n ? kq[e] : KEY_NONE);
            }
        }
    }
}

hipError_t launch_band_select(const SearchArgs& a, const float* S, size_t ldS, int sub, int max_slots, hipStream_t s) {
    if (max_slots <= 0 || a.n_docs <= 0) return hipSuccess;
    if (a.dim % 4 || a.dim > 64 * 4 * MERGE_MAXV || a.k > BAND_KQ || !a.flag_count || !a.flag_list || !a.flag_tau || !a.flag_top ||
        !a.flag2_count || !a.flag2_list)
        return hipErrorInvalidValue;
    static unsigned long long attr = 0;     // bit d: set on device d
    set_max_dynamic_lds((const void*)band_select_kernel, (int)sizeof(BandLds), attr);
    // a fixed grid walks the groups of flagged slots (one workgroup per CU: 144 KiB of LDS); all leave at once when nothing is flagged
    const int groups = (max_slots + BAND_G - 1) / BAND_G;
    hipLaunchKernelGGL(band_select_kernel, dim3(groups < 256 ? groups : 256), dim3(BAND_NT), sizeof(BandLds), s, a, S, ldS, sub, max_slots);
    return hipGetLastError();
}

}  // namespace vr
