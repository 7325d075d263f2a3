// Fused similarity GEMM + wavefront bitonic top-k over the HBM-resident index.
// Replaces torch.matmul(Q, C^T) + torch.topk (dense_retriever.py:28-30): the Nq x Nd score
// matrix is never written.
//
//  sweep   one workgroup = (doc chunk, 128-query tile).  For every 128-doc tile of its chunk it
//          runs the bf16 MFMA main loop of gemm_core.h (A = index rows, W = query rows) and
//          filters the 128x128 scores in registers against a per-query running threshold kept
//          in LDS; survivors are appended to a 64-entry per-query LDS buffer, and a buffer that
//          passes 32 entries is compacted by ONE wave with a 64-lane bitonic sort on
//          (score desc, doc id asc) keys — one candidate per lane, cross-lane exchange only.
//          Result: the exact bf16-score top-KP of the chunk per query.
//  merge   one workgroup per query: the per-chunk lists -> the best gathered bf16 candidates, the
//          top KP re-scored against the fp32 index (exact fp32 dot products), then CERTIFIED
//          (search_common.h: certify_tail): every further candidate whose bf16 score could still
//          reach the fp32 top-k under the bf16 error bound is re-scored too, and a query whose
//          lists cannot prove completeness goes to the exact fp32 pass (search_exact.hip).  The
//          ids returned are the fp32 ranking's, not a tolerance-equivalent of it.
// Block placement: the 8 query tiles of a chunk are consecutive on ONE XCD (b % 8 == chunk % 8)
// so the index tile is fetched from HBM once and re-read from that XCD's L2.
// Roofline: MFMA for Nq >~ 300 (2*Nq*Nd*D flop), HBM for small Nq (Nd*D*2 bytes per sweep).
#include <cstdlib>

#include "gemm_core.h"
#include "search_band.h"
#include "search_common.h"
#include "kernels.h"

namespace vr {

struct SweepLds {
    float thr[128];
    int cnt[128];
    uint64_t cand[128][SRCH_CAP];
};
constexpr int SWEEP_SMEM = GEMM_SMEM_BYTES + (int)sizeof(SweepLds);

// tile_step > 1 is the threshold PRE-PASS: each chunk visits one tile, tiles spread evenly over
// the index; its per-query KP-th best score (search_thr_kernel) is a valid lower bound of the
// global KP-th best and seeds the thresholds of the main sweep, which then rejects ~98 % of the
// scores in registers from its first tile on (no per-chunk warm-up of sorts).
template <int KP>
__global__ __launch_bounds__(256) void search_sweep_kernel(SearchArgs p, int q_tiles, int tiles_per_chunk,
                                                           int tile_step, const float* __restrict__ thr_init) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SweepLds& L = *reinterpret_cast<SweepLds*>(smem + GEMM_SMEM_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;

    const int b = blockIdx.x;
    const int chunk = (b / (8 * q_tiles)) * 8 + (b & 7);
    const int qt = (b >> 3) % q_tiles;
    const int q0 = qt * 128;
    const int n_tiles = (int)((p.n_docs + 127) / 128);
    const int tile_lo = chunk * tiles_per_chunk;
    const int tile_hi = min((n_tiles + tile_step - 1) / tile_step, tile_lo + tiles_per_chunk);

    if (tid < 128) {
        // padding queries (zero rows, every score ties at 0) must never collect candidates
        float t0 = thr_init ? thr_init[q0 + tid] : -INFINITY;
        if (q0 + tid >= p.nq) t0 = INFINITY;
        L.thr[tid] = t0;
        L.cnt[tid] = 0;
    }
    __syncthreads();

    // one wave compacts the buffers of its 32 queries: sort, keep the best KP, raise thr
    auto compact = [&](bool force) {
        const int qn = wave * 32 + (lane & 31);
        const int c_l = L.cnt[qn];
        unsigned long long todo = __ballot((lane < 32) && (force ? c_l > 0 : c_l > SRCH_TRIG));
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int qq = wave * 32 + src;
            const int c = __shfl(c_l, src, 64);
            uint64_t key = (lane < c) ? L.cand[qq][lane] : KEY_NONE;
            key = wave_bitonic_desc(key, lane);
            if (lane < KP) L.cand[qq][lane] = key;
            const int keep = min(c, KP);
            if (lane == KP - 1 && c >= KP) L.thr[qq] = orderable_f32((uint32_t)(key >> 32));
            if (lane == 0) L.cnt[qq] = keep;
        }
    };

    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        const int doc0 = tile * tile_step * 128;
        gemm_acc_t acc;
        gemm_zero(acc);
        gemm_mainloop(acc, (const bf16_t*)p.index_bf16, p.dim, (const bf16_t*)p.q_bf16, p.dim,
                            doc0, q0, p.dim, smem);
        // thresholds of this lane's 16 queries (4 fragments x 4 consecutive columns)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int doc = doc0 + wm * 64 + i * 16 + fr;
            bool any = false;
            if (doc < p.n_docs) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int qn = wn * 64 + j * 16 + fq * 4;
                    const f32x4 th = *reinterpret_cast<const f32x4*>(&L.thr[qn]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float s = acc[i][j][r];
                        if (s >= th[r]) {
                            const int pos = atomicAdd(&L.cnt[qn + r], 1);
                            L.cand[qn + r][pos] = make_key(s, (uint32_t)doc);
                            any = true;
                        }
                    }
                }
            }
            if (__syncthreads_or(any)) {     // rare after warm-up: thresholds reject almost all
                compact(false);
                __syncthreads();
            }
        }
    }
    __syncthreads();
    compact(true);
    __syncthreads();
    // emit [query][chunk][KP] (score, id); unused slots: -inf / -1
    for (int e = tid; e < 128 * KP; e += 256) {
        const int qq = e / KP, s = e % KP;
        const size_t o = ((size_t)(q0 + qq) * p.n_chunks + chunk) * KP + s;
        if (s < L.cnt[qq]) {
            const uint64_t key = L.cand[qq][s];
            p.cand_scores[o] = orderable_f32((uint32_t)(key >> 32));
            p.cand_ids[o] = (int)(~(uint32_t)key);
        } else {
            p.cand_scores[o] = -INFINITY;
            p.cand_ids[o] = -1;
        }
    }
}

// ---- threshold pre-pass ----------------------------------------------------------------------
// One workgroup = one sampled 128-doc tile x 128 queries: the plain GEMM tile, no candidate
// machinery.  A lane holds 4 docs (its 4 row fragments) of each of its 16 query columns; it emits
// the max of the 4, so a query gets 32 group maxima per tile, each over 4 DISTINCT docs.
// search_thr_kernel folds the PRE_CHUNKS*32 = 1024 maxima of a query to 64 (lane-local max of 16)
// and sorts them once: the KP-th largest is <= the scores of KP distinct docs, i.e. a valid lower
// bound of the query's global KP-th best score, and nearly as tight as the KP-th best of the
// 4096-doc sample (the top KP rarely share a 64-doc group).  32 sampled tiles x the 128-query tiles
// still fit one round of workgroups (256 for 1000 queries), so the wider sample costs no time, and
// the sweep's filter lets ~450 instead of ~1900 candidates per query through (1k x 100k: 0.567 ->
// 0.536 ms with 8 -> 32 tiles, 0.547 with 64).
constexpr int PRE_CHUNKS = 32;     // sampled 128-row tiles = 4096 rows at most (8 or 16 for small shards: one tile in eight)

// The pre-pass's K loop: gemm_core.h's 128 x 128 x 64 tile with FOUR LDS stages instead of two.  The sampled index rows come
// from HBM, cold, one workgroup per CU: with two stages (the load of step t + 1 issued when step t starts, everything drained at
// the next barrier) a K-step was one memory round trip — 36 steps x ~1 us = 36 us for 0.28 us of MFMAs each.  Here the loads run
// three steps ahead behind a COUNTED wait (a wave's eight 1 KiB LDS-DMA loads per step complete in order: `vmcnt(16)` = step kt
// has landed, the two younger steps stay in flight) and a raw barrier (`__syncthreads()` would drain them: its fence waits for
// `vmcnt(0)` while an LDS-DMA is pending).  The barrier of step kt also says every wave is done with step kt - 1: its stage is
// the one the loads of step kt + 3 go to.
constexpr int PRE_NST = 4;
constexpr int PRE_SMEM_BYTES = PRE_NST * 2 * GEMM_TILE_BYTES;      // 128 KiB
__device__ __forceinline__ void prepass_mainloop(gemm_acc_t& acc, const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W,
                                                 int ldw, int m0, int n0, int K, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = K / GEMM_BK;
    auto stage = [&](int kt) {
        char* st = smem + (kt % PRE_NST) * 2 * GEMM_TILE_BYTES;
        stage_glds(A, lda, m0, kt * GEMM_BK, st, wave, lane);
        stage_glds(W, ldw, n0, kt * GEMM_BK, st + GEMM_TILE_BYTES, wave, lane);
    };
    for (int t = 0; t < PRE_NST - 1 && t < nk; ++t) stage(t);
    for (int kt = 0; kt < nk; ++kt) {
        const int younger = min(PRE_NST - 1, nk - kt) - 1;             // steps behind kt that are in flight (8 loads per wave each)
        if (younger >= 2) VR_WAIT_VM_BARRIER(16);
        else if (younger == 1) VR_WAIT_VM_BARRIER(8);
        else VR_WAIT_VM_BARRIER(0);
        if (kt + PRE_NST - 1 < nk) stage(kt + PRE_NST - 1);
        const char* cur = smem + (kt % PRE_NST) * 2 * GEMM_TILE_BYTES;
        gemm_compute_tile(acc, cur, cur + GEMM_TILE_BYTES, wm, wn, lane);
    }
    __syncthreads();
}
constexpr int PRE_GROUPS = 32;     // group maxima per (query, tile)

__global__ __launch_bounds__(256) void search_prepass_kernel(SearchArgs p, int q_tiles, int tile_step, int pre_chunks,
                                                             float* __restrict__ gmax) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int chunk = blockIdx.x / q_tiles, q0 = (blockIdx.x % q_tiles) * 128;
    const int doc0 = chunk * tile_step * 128;
    gemm_acc_t acc;
    gemm_zero(acc);
    prepass_mainloop(acc, (const bf16_t*)p.index_bf16, p.dim, (const bf16_t*)p.q_bf16, p.dim, doc0, q0, p.dim, smem);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (doc0 + wm * 64 + i * 16 + fr < p.n_docs) m = fmaxf(m, acc[i][j][r]);
            const int qn = q0 + wn * 64 + j * 16 + fq * 4 + r;
            gmax[((size_t)qn * pre_chunks + chunk) * PRE_GROUPS + wm * 16 + fr] = m;
        }
    }
}

template <int KP>
__global__ __launch_bounds__(256) void search_thr_kernel(const float* __restrict__ gmax, int nq_pad, int pre_chunks,
                                                         float* __restrict__ thr) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq_pad) return;
    const float* g = gmax + (size_t)q * pre_chunks * PRE_GROUPS;
    float m = -INFINITY;
    for (int t = 0; t < pre_chunks * PRE_GROUPS / 64; ++t) m = fmaxf(m, g[t * 64 + lane]);
    const uint64_t sorted = wave_sort_desc((uint64_t)f32_orderable(m) << 32);
    const uint64_t kth = shfl_u64(sorted, KP - 1);
    if (lane == 0) thr[q] = orderable_f32((uint32_t)(kth >> 32));
}


// ---- the pre-pass that OWNS its sample (round 6; the 256-tile sweep on the one-wave kernel, >= 128 index tiles) -----------
// The plain pre-pass scores 4096 sampled rows and throws the scores away: the sweep scores them again.  Here the sample is
// SEARCH_PRE_SPOTS = 16 whole 256-row index tiles (tile k * S + S - 1, S = tiles / 16: the same 4096 rows spread over the index),
// every score is kept ([query][4096] fp32 behind the group maxima), and search_thr_own_kernel — after the threshold — appends
// the sampled rows that reach it to the query's OWN lists: the last pre_own_chunks list chunks, 16 half-lists = 1024 slots, in
// the format of the sweep's.  The sweep walks the other tiles only.  1k x 100k: 391 - 16 = 375 tiles over 64 chunks — six tiles
// per workgroup instead of seven.  Rows >= thr sit in fold groups (64 rows) whose maximum is >= thr: the KP - 1 groups above the
// KP-th maximum plus the groups that tie with it — ~18 rows on model embeddings, at most 1024 for KP = 16 without exact ties.  More
// than 1024 (KP = 32 in its worst case, duplicate rows): the query's lists are declared incomplete (thr_cert = +inf: the merge
// cannot certify, the band pass redoes the query) — rigorous, just slow.
__global__ __launch_bounds__(256) void search_prepass_own_kernel(SearchArgs p, int q_tiles, int pre_step, float* __restrict__ gmax,
                                                                 float* __restrict__ gscore) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int pre_chunks = 2 * SEARCH_PRE_SPOTS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
    const int chunk = blockIdx.x / q_tiles, q0 = (blockIdx.x % q_tiles) * 128;
    const int doc0 = ((chunk >> 1) * pre_step + pre_step - 1) * 256 + (chunk & 1) * 128;
    gemm_acc_t acc;
    gemm_zero(acc);
    prepass_mainloop(acc, (const bf16_t*)p.index_bf16, p.dim, (const bf16_t*)p.q_bf16, p.dim, doc0, q0, p.dim, smem);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qn = q0 + wn * 64 + j * 16 + fq * 4 + r;
            // (a lane's four rows i * 16 + fr of the 64-row block go to positions fr * 4 + i: ONE 16-byte store, 256 contiguous bytes
            // per 16 lanes — search_thr_own_kernel undoes the permutation when it names the row)
            float* gs = gscore + (size_t)qn * (pre_chunks * 128) + chunk * 128 + wm * 64 + fr * 4;
            float m = -INFINITY;
            f32x4 v4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v4[i] = doc0 + wm * 64 + i * 16 + fr < p.n_docs ? acc[i][j][r] : -INFINITY;
                m = fmaxf(m, v4[i]);
            }
            *reinterpret_cast<f32x4*>(gs) = v4;
            gmax[((size_t)qn * pre_chunks + chunk) * PRE_GROUPS + wm * 16 + fr] = m;
        }
    }
}

// one wave per query: threshold as in search_thr_kernel, then the query's 4096 sampled scores against it
template <int KP>
__global__ __launch_bounds__(256) void search_thr_own_kernel(SearchArgs p, const float* __restrict__ gmax, const float* __restrict__ gscore,
                                                             int nq_pad, int pre_step, float* __restrict__ thr, float* __restrict__ thr_cert) {
    constexpr int pre_chunks = 2 * SEARCH_PRE_SPOTS, NS = pre_chunks * 128, CAP = 64;
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq_pad) return;
    const float* g = gmax + (size_t)q * pre_chunks * PRE_GROUPS;
    // the query's 4096 scores are requested FIRST: they do not depend on the threshold, and the fold + sort below then run under
    // their latency instead of in front of it (one memory round trip per wave instead of two)
    f32x4 v[NS / 256];
    {
        const float* gs = gscore + (size_t)q * NS;
#pragma unroll
        for (int t = 0; t < NS / 256; ++t) v[t] = *reinterpret_cast<const f32x4*>(gs + t * 256 + lane * 4);
    }
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < pre_chunks * PRE_GROUPS / 64; ++t) m = fmaxf(m, g[t * 64 + lane]);
    const uint64_t sorted = wave_sort_desc((uint64_t)f32_orderable(m) << 32);
    const uint64_t kth = shfl_u64(sorted, KP - 1);
    const float th = orderable_f32((uint32_t)(kth >> 32));
    const int own_lists = p.pre_own_chunks * 2, slots = own_lists * CAP;
    const size_t list0 = ((size_t)q * p.n_chunks + (p.n_chunks - p.pre_own_chunks)) * 2;     // first own half-list of the query
    unsigned long long* keys = p.cand_keys + list0 * CAP;
    int base = 0;
    if (q < p.nq) {                                      // (padding queries: empty lists, threshold +inf in the sweep anyway)
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int t = 0; t < NS / 256; ++t) {
            const float mx = fmaxf(fmaxf(v[t][0], v[t][1]), fmaxf(v[t][2], v[t][3]));
            if (__ballot(mx >= th && mx > -INFINITY) == 0ull) continue;      // (~18 of 4096 rows reach the threshold)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool pass = v[t][u] >= th && v[t][u] > -INFINITY;
                const unsigned long long bal = __ballot(pass);
                if (bal) {
                    if (pass) {
                        const int pos = base + __popcll(bal & lt);
                        const int e = t * 256 + lane * 4 + u, c = e >> 7, w = e & 63;       // position fr * 4 + i of its 64-row block
                        const int doc = ((c >> 1) * pre_step + pre_step - 1) * 256 + (c & 1) * 128 + (e & 64) + (w & 3) * 16 + (w >> 2);
                        if (pos < slots) keys[(pos % own_lists) * CAP + pos / own_lists] = make_key(v[t][u], (uint32_t)doc);
                    }
                    base += __popcll(bal);
                }
            }
        }
    }
    // (dealt round-robin over the own half-lists: the merge walks a list with ONE thread — eighteen entries in one list were a
    // chain of five dependent cold loads behind everybody else's one, +8 us per merge)
    if (lane < own_lists) p.cand_ids[list0 + lane] = min(CAP, (base + own_lists - 1 - lane) / own_lists);
    if (lane == 0) {
        thr[q] = th;
        thr_cert[q] = base > slots ? INFINITY : th;     // overflow: rows were dropped, the lists prove nothing
    }
}


// ---- merge of the sorted chunk lists (128-tile sweep, streaming kernel): one workgroup per query ----
// 256 threads read all chunk entries with independent loads, bound the GD-th best from the chunk
// heads (GD = KP + 16: the certification below may want candidates past the KP-th), filter, sort the
// survivors once, and the four waves re-score in parallel (certify_tail).
// What the lists do NOT hold: a full list (KP entries) may have dropped rows below its last entry —
// dropB is the largest such last entry; a shorter list holds every row of its chunk that passed the
// sweep's starting threshold.
// NT threads: 1024 for the streaming kernel's handful of queries — the chip is empty next to these <= 16 workgroups, and
// the fp32 re-scoring of a query's 16..40 candidate rows (9 KB each, a memory round trip per row and wave) is the longest
// chain of the whole search: sixteen waves take it in one or two rounds instead of four to ten.
// With the sweep's score rows at hand (SearchArgs::score_rows: the streaming kernel writes them) a query that cannot be
// certified is redone HERE, by the same workgroup (search_band.h; dynamic LDS: BandLds) — nothing is launched behind the merge.
template <int KP, int NT>
__global__ __launch_bounds__(NT) void search_merge_wg_kernel(SearchArgs p) {
    extern __shared__ __attribute__((aligned(16))) char band_smem[];
    constexpr int GD = KP + MERGE_GD_EXTRA < 64 ? KP + MERGE_GD_EXTRA : 64;
    __shared__ uint64_t lm[NT];
    __shared__ uint64_t surv[MERGE_CAP], exact_w[MERGE_CAP];
    __shared__ uint64_t cand[64], exact_s[64];
    __shared__ uint64_t thr_s;
    __shared__ unsigned drop_s;
    __shared__ int n_s, x_s;
    __shared__ float tau_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x;
    const int total = p.n_chunks * KP;
    const float* cs = p.cand_scores + (size_t)q * total;
    const int* ci = p.cand_ids + (size_t)q * total;
    // bound from the chunk HEADS (lists are sorted, so a head is its chunk's best): thread-local max
    // over its chunks; the GD-th largest of the 64 folded maxima is <= GD distinct heads.  (Folding
    // arbitrary entries instead mixes list positions and gives a far looser bound.)
    uint64_t m = KEY_NONE;
    unsigned drop = 0u;                                  // orderable score of the best LAST entry of a full list
    if (tid == 0) drop_s = 0u;
    for (int c = tid; c < p.n_chunks; c += NT) {
        const int id = ci[c * KP];
        const float sc = cs[c * KP];
        const uint64_t key = id >= 0 ? make_key(sc, (uint32_t)id) : KEY_NONE;
        m = key > m ? key : m;
        if (ci[c * KP + KP - 1] >= 0) drop = max(drop, f32_orderable(cs[c * KP + KP - 1]));
    }
    lm[tid] = m;
    __syncthreads();
    if (drop) atomicMax(&drop_s, drop);
    if (wave == 0) {
        uint64_t v = lm[lane];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) { const uint64_t o = lm[lane + 64 * w]; v = o > v ? o : v; }
        const uint64_t t = shfl_u64(wave_sort_desc(v), GD - 1);
        if (lane == 0) thr_s = t;
    }
    __syncthreads();
    // every entry with key >= thr -> surv (<= MERGE_CAP of them), the best 64 sorted into cand; returns their number
    auto gather = [&](uint64_t thr) -> int {
        if (tid == 0) n_s = 0;
        __syncthreads();
        for (int e0 = tid; e0 < total; e0 += NT * 8) {
            int id8[8];
            float sc8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = min(e0 + NT * u, total - 1);
                id8[u] = (e0 + NT * u < total) ? ci[e] : -1;
                sc8[u] = cs[e];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (id8[u] >= 0) {
                    const uint64_t key = make_key(sc8[u], (uint32_t)id8[u]);
                    if (key >= thr) { const int pos = atomicAdd(&n_s, 1); if (pos < MERGE_CAP) surv[pos] = key; }
                }
            }
        }
        __syncthreads();
        const int n = n_s;
        if (wave == 0) {
            uint64_t best = KEY_NONE;
            if (n <= MERGE_CAP) {
                for (int base = 0; base < n; base += 64) {
                    const uint64_t key = (base + lane < n) ? surv[base + lane] : KEY_NONE;
                    best = (base == 0) ? wave_sort_desc(key) : wave_merge_top64(best, key, lane);
                }
            } else {                                        // massive ties: merge everything
                for (int base = 0; base < total; base += 64) {
                    const int e = base + lane;
                    uint64_t key = KEY_NONE;
                    if (e < total && ci[e] >= 0) key = make_key(cs[e], (uint32_t)ci[e]);
                    best = (base == 0) ? wave_sort_desc(key) : wave_merge_top64(best, key, lane);
                }
            }
            cand[lane] = best;
        }
        __syncthreads();
        return n;
    };
    const uint64_t thr = thr_s;
    const int n = gather(thr);
    // entries outside `cand`: below the gather bound, or (more than 64 gathered) below cand[63]
    const float coverB = n > 64 ? key_score(cand[63]) : (thr == KEY_NONE ? -INFINITY : key_score(thr));
    float dropB = drop_s ? orderable_f32(drop_s) : -INFINITY;
    if (p.thr_used) dropB = fmaxf(dropB, p.thr_used[q]);
    const bool in_place = p.score_rows != nullptr;
    const bool flagged = certify_tail<KP>(p, q, cand, exact_s, coverB, dropB, &tau_s, &x_s,
                                          [&](float tau) { return gather((uint64_t)f32_orderable(tau) << 32); }, surv, exact_w, in_place);
    if (in_place && flagged) {                           // workgroup-uniform
        __syncthreads();
        const float tau = tau_s;
        band_pass_in_place<NT>(p, q, tau, p.score_rows + (size_t)q * p.ld_scores, *reinterpret_cast<BandLds*>(band_smem), &x_s);
    }
}

int search_kprime(int k) {
    if (k <= 0) return 0;
    if (k <= 10) return 16;
    if (k <= 26) return 32;
    return 0;
}

int search_prepass_floats() { return PRE_CHUNKS * PRE_GROUPS + 2 * SEARCH_PRE_SPOTS * 128; }   // group maxima + (owning form) every sampled score

bool search_uses_256(int nq) {
    // 17..128 queries also run faster on the 256^2 sweep (half-empty query tile and all: 0.29 vs 0.33 ms)
    return nq >= 17;
}

int search_num_chunks(int64_t n_docs, int nq) {
    if (search_uses_256(nq)) {       // one 8-wave workgroup per CU: ~256 workgroups
        const int q_tiles = (nq + 255) / 256;
        const int n_tiles = (int)((n_docs + 255) / 256);
        int chunks = ((256 / q_tiles) + 7) / 8 * 8;
        if (chunks < 8) chunks = 8;
        // a small shard: the smallest multiple of 8 that gives every index tile its own chunk (the chunks past the last tile are
        // empty workgroups).  Rounding DOWN — 48 chunks for the 49 tiles of an 8-way shard of the 100k index — left one chunk
        // with two tiles and the whole sweep waiting for it: 116 us instead of 58 (round 5, tools/search_stages.py).
        while (chunks > 8 && chunks - 8 >= n_tiles) chunks -= 8;
        return chunks;
    }
    const int q_tiles = (nq + 127) / 128;
    const int n_tiles = (int)((n_docs + 127) / 128);
    int chunks = ((512 / q_tiles) + 7) / 8 * 8;          // ~2 workgroups per CU
    if (chunks < 8) chunks = 8;
    while (chunks > 8 && chunks > n_tiles) chunks -= 8;
    return chunks;
}

constexpr int SMALL_NQ = 16;        // up to here: no pre-pass, workgroup-per-query merge

#ifndef VR_PREPASS_OWN
#define VR_PREPASS_OWN 1
#endif
int search_prepass_owned(int64_t n_docs, int nq, int dim) {
    const int64_t tiles = (n_docs + 255) / 256;
    // (one tile in eight at most — as for the plain pre-pass — and the one-wave sweep's conditions: search256w.hip)
    if (!VR_PREPASS_OWN || !search_uses_256(nq) || nq <= SMALL_NQ || dim % 128 || tiles < 8 * SEARCH_PRE_SPOTS) return 0;
    // only where it shortens the sweep: the lists of its own cost the threshold kernel and the merge ~15 us (1k x 100k: 7 -> 6
    // tiles per workgroup, -36 us of sweep; 256 queries x 100k: 2 tiles per workgroup either way)
    const int c_all = search_num_chunks(n_docs, nq), c_own = search_num_chunks(n_docs - (int64_t)SEARCH_PRE_SPOTS * 256, nq);
    const int64_t tpc_all = (tiles + c_all - 1) / c_all, tpc_own = (tiles - SEARCH_PRE_SPOTS + c_own - 1) / c_own;
    return tpc_own < tpc_all ? 8 : 0;
}

template <int KP>
static hipError_t launch_kp(const SearchArgs& a, hipStream_t s) {
    const int q_tiles = (a.nq + 127) / 128;
    const int n_tiles = (int)((a.n_docs + 127) / 128);
    auto k = search_sweep_kernel<KP>;
    static unsigned long long attr = 0;     // bit d: set on device d
    set_max_dynamic_lds((const void*)k, SWEEP_SMEM, attr);
    hipError_t e;
    const float* thr = nullptr;
    // (a handful of queries: the chunks warm their thresholds up themselves — cheaper than two more launches)
    // sampled tiles: 32, or fewer on small shards (never more than one tile in eight: the pre-pass must stay small
    // against the sweep — an 8-way shard of the 100k index has 98 tiles)
    int pre_chunks = PRE_CHUNKS;
    while (pre_chunks > 8 && n_tiles < 8 * pre_chunks) pre_chunks /= 2;
    const float* thr_cert = nullptr;
    if (a.pre_own_chunks) {
        // the pre-pass owns its sample (see search_prepass_own_kernel): the caller sized the lists with search_prepass_owned
        if (!a.thr_init || !a.thr_cert || !a.cand_keys || a.pre_own_chunks != search_prepass_owned(a.n_docs, a.nq, a.dim) ||
            a.n_chunks <= a.pre_own_chunks)
            return hipErrorInvalidValue;
        static unsigned long long pattr = 0;    // bit d: set on device d
        set_max_dynamic_lds((const void*)search_prepass_own_kernel, PRE_SMEM_BYTES, pattr);
        const int pre_step = (int)((a.n_docs + 255) / 256) / SEARCH_PRE_SPOTS;
        float* gmax = a.cand_scores;                                   // [nq_pad128][32][PRE_GROUPS]
        float* gscore = gmax + (size_t)q_tiles * 128 * PRE_CHUNKS * PRE_GROUPS;    // [nq_pad128][4096]
        static_assert(PRE_CHUNKS == 2 * SEARCH_PRE_SPOTS, "32 sampled 128-row tiles = 16 index tiles of 256 rows");
        hipLaunchKernelGGL(search_prepass_own_kernel, dim3(PRE_CHUNKS * q_tiles), dim3(256), PRE_SMEM_BYTES, s, a, q_tiles, pre_step,
                           gmax, gscore);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        hipLaunchKernelGGL(search_thr_own_kernel<KP>, dim3(q_tiles * 32), dim3(256), 0, s, a, (const float*)gmax, (const float*)gscore,
                           q_tiles * 128, pre_step, a.thr_init, a.thr_cert);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        thr = a.thr_init;
        thr_cert = a.thr_cert;
    } else if (a.thr_init && n_tiles >= 8 * pre_chunks && a.nq > SMALL_NQ) {
        static unsigned long long pattr = 0;    // bit d: set on device d
        set_max_dynamic_lds((const void*)search_prepass_kernel, PRE_SMEM_BYTES, pattr);
        float* gmax = a.cand_scores;                   // [nq_pad128][pre_chunks][PRE_GROUPS], dead before the sweep writes
        hipLaunchKernelGGL(search_prepass_kernel, dim3(pre_chunks * q_tiles), dim3(256), PRE_SMEM_BYTES, s, a, q_tiles,
                           n_tiles / pre_chunks, pre_chunks, gmax);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        hipLaunchKernelGGL(search_thr_kernel<KP>, dim3(q_tiles * 32), dim3(256), 0, s, (const float*)gmax, q_tiles * 128,
                           pre_chunks, a.thr_init);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        thr = a.thr_init;
    }
    if (a.prof_ev && (e = hipEventRecord(a.prof_ev[2], s)) != hipSuccess) return e;
    SearchArgs am = a;
    am.thr_used = thr_cert ? thr_cert : thr;                                 // what the lists are complete down to
    if (search_uses_256(a.nq)) return launch_sweep256(am, KP, thr, s);      // sweep + its own merge
    if (search_uses_stream(a.nq, a.dim)) {
        if ((e = launch_search_stream(a, KP, s)) != hipSuccess) return e;
    } else {
        const int tpc = (n_tiles + a.n_chunks - 1) / a.n_chunks;
        hipLaunchKernelGGL(k, dim3(a.n_chunks * q_tiles), dim3(256), SWEEP_SMEM, s, a, q_tiles, tpc, 1, thr);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (a.prof_ev && (e = hipEventRecord(a.prof_ev[3], s)) != hipSuccess) return e;
    if (search_uses_stream(a.nq, a.dim)) {
        auto km = search_merge_wg_kernel<KP, 1024>;
        const int lds = am.score_rows ? (int)sizeof(BandLds) : 0;
        static unsigned long long mattr = 0;    // bit d: set on device d
        set_max_dynamic_lds((const void*)km, (int)sizeof(BandLds), mattr);
        hipLaunchKernelGGL(km, dim3(a.nq), dim3(1024), lds, s, am);
    } else {
        am.score_rows = nullptr;
        hipLaunchKernelGGL((search_merge_wg_kernel<KP, 256>), dim3(a.nq), dim3(256), 0, s, am);
    }
    return hipGetLastError();
}

hipError_t launch_search(const SearchArgs& a, hipStream_t s) {
    if (a.nq <= 0) return hipSuccess;
    if (a.dim % 64 || a.dim > 64 * 4 * MERGE_MAXV || a.n_chunks % 8) return hipErrorInvalidValue;
    switch (search_kprime(a.k)) {
        case 16: return launch_kp<16>(a, s);
        case 32: return launch_kp<32>(a, s);
        default: return hipErrorInvalidValue;
    }
}

// ---- multi-GPU: merge per-shard lists after the all-gather ------------------------------------
// input either (score, global id) pairs or packed keys (SearchArgs::out_keys format, global ids inside)
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ scores,
                                                         const int64_t* __restrict__ ids,
                                                         const unsigned long long* __restrict__ keys, int n_parts,
                                                         int nq, int k, float* __restrict__ out_scores,
                                                         int64_t* __restrict__ out_ids) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const int total = n_parts * k;
    uint64_t best = KEY_NONE;
    for (int base = 0; base < total; base += 64) {
        const int e = base + lane;
        uint64_t key = KEY_NONE;
        if (e < total) {
            const int part = e / k, s = e % k;
            const size_t o = ((size_t)part * nq + q) * k + s;
            if (keys) key = keys[o];
            else if (ids[o] >= 0) key = make_key(scores[o], (uint32_t)ids[o]);
        }
        best = (base == 0) ? wave_bitonic_desc(key, lane) : wave_merge_top64(best, key, lane);
    }
    if (lane < k) {
        const bool ok = best != KEY_NONE;
        out_scores[(size_t)q * k + lane] = ok ? orderable_f32((uint32_t)(best >> 32)) : -INFINITY;
        out_ids[(size_t)q * k + lane] = ok ? (int64_t)(~(uint32_t)best) : (int64_t)-1;
    }
}

hipError_t launch_topk_merge_any(const float* scores, const int64_t* ids, const unsigned long long* keys, int n_parts,
                                 int nq, int k, float* out_scores, int64_t* out_ids, hipStream_t s);   // search_bigk.hip

hipError_t launch_topk_merge(const float* scores, const int64_t* ids, int n_parts, int nq, int k,
                             float* out_scores, int64_t* out_ids, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    if (k <= 0) return hipErrorInvalidValue;
    if (k > 64) return launch_topk_merge_any(scores, ids, nullptr, n_parts, nq, k, out_scores, out_ids, s);
    hipLaunchKernelGGL(topk_merge_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, scores, ids, (const unsigned long long*)nullptr,
                       n_parts, nq, k, out_scores, out_ids);
    return hipGetLastError();
}

hipError_t launch_topk_merge_keys(const unsigned long long* keys, int n_parts, int nq, int k, float* out_scores,
                                  int64_t* out_ids, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    if (k <= 0) return hipErrorInvalidValue;
    if (k > 64) return launch_topk_merge_any(nullptr, nullptr, keys, n_parts, nq, k, out_scores, out_ids, s);
    hipLaunchKernelGGL(topk_merge_kernel, dim3((nq + 3) / 4), dim3(256), 0, s, (const float*)nullptr, (const int64_t*)nullptr,
                       keys, n_parts, nq, k, out_scores, out_ids);
    return hipGetLastError();
}

}  // namespace vr
