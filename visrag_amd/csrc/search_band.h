// The band pass of ONE flagged query by ONE workgroup (search_band.hip has the why): every index row whose bf16-MFMA score
// S[row] is >= tau is re-scored in fp32, the keys reduced to the top k.  Shared by band_select_kernel (a workgroup of eight
// waves per flagged query behind the band GEMM) and — round 6 — by the merge kernel of the handful-of-queries search itself:
// the streaming sweep has every score in registers anyway and writes them out (6.4 MB next to the 460 MB it reads), so a
// flagged query of that path is redone by ITS OWN merge workgroup: no band GEMM, no selection launch, no exact-pass launches —
// the four launches that used to follow every search and leave at once (11.6 us of a 117 us single query) are gone.
// (A band beyond BAND_MAX rows is the exception: see band_pass_in_place — walked here once, the exact pass from then on.)
#pragma once
#include "kernels.h"
#include "search_common.h"

namespace vr {

constexpr int BAND_MAX = 8192;      // rows a query's band may hold per pass (its keys: 64 KiB of LDS)
constexpr int BAND_NR = 3;          // rows in flight per wave

struct BandLds {                    // dynamic LDS image: 96 KiB, one workgroup per CU
    uint64_t keys[BAND_MAX];        // the band's exact keys
    uint32_t cand[BAND_MAX];        // its row ids
};

// Rows [lo, hi) of the score row: gather the band (S >= tau) into L.cand, returns its size (may exceed BAND_MAX: then only
// the first BAND_MAX positions were stored and the caller must not use them).  16 bytes per thread and load, four loads
// requested before the first is looked at (the ballots keep the compiler from hoisting them itself: without this a lone
// workgroup pays a memory round trip per 8 KiB of scores).  Called by the whole workgroup; ucnt_s: LDS scratch.
template <int NT>
__device__ __forceinline__ int band_gather(const float* __restrict__ row, int lo, int hi, float tau, BandLds& L, int* ucnt_s) {
    const int tid = threadIdx.x, lane = tid & 63;
    __syncthreads();                                                     // (LDS of the previous use is free)
    if (tid == 0) *ucnt_s = 0;
    __syncthreads();
    for (int i0 = lo; i0 < hi; i0 += 16 * NT) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 4 * NT + tid * 4;
            v[u] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (i < hi) v[u] = *reinterpret_cast<const f32x4*>(row + i);                  // (rows are padded to 256 floats; lo, hi % 4 == 0 or hi = n_docs)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * 4 * NT + tid * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool in = i + e < hi && v[u][e] >= tau;
                const unsigned long long b = __ballot(in);
                if (b == 0ull) continue;                                 // wave-uniform
                int base = 0;
                if (lane == 0) base = atomicAdd(ucnt_s, __popcll(b));
                base = __shfl(base, 0, 64);
                const int pos = base + __popcll(b & ((1ull << lane) - 1ull));
                if (in && pos < BAND_MAX) L.cand[pos] = (uint32_t)(i + e);
            }
        }
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(*ucnt_s);
}

// exact fp32 scores of the n_band <= BAND_MAX gathered rows -> L.keys: a wave takes rows wave, wave + NW, ...; BAND_NR rows in
// flight per wave (a row is nine 1 KiB loads; the re-scoring is a chain of memory round trips)
template <int NT>
__device__ __forceinline__ void band_rescore(const SearchArgs& p, const f32x4 (&qv)[MERGE_MAXV], int n_band, BandLds& L) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nv = p.dim >> 2;
    auto load_row = [&](f32x4 (&dv)[MERGE_MAXV], int cidx) {
        if (cidx >= n_band) return;
        const f32x4* dr = reinterpret_cast<const f32x4*>(p.index_f32 + (size_t)L.cand[cidx] * p.dim);
#pragma unroll
        for (int i = 0; i < MERGE_MAXV; ++i) {
            const int cc = lane + i * 64;
            dv[i] = (cc < nv) ? dr[cc] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto score_row = [&](const f32x4 (&dv)[MERGE_MAXV], int cidx) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < MERGE_MAXV; ++i) {
            const int cc = lane + i * 64;
            if (cc < nv) a = dot_chunk(qv[i], dv[i], a);                  // the chain of dot_lane(q, row)
        }
        a = wave_sum(a);
        if (lane == 0) L.keys[cidx] = make_key(a, L.cand[cidx]);
    };
    f32x4 dr_[BAND_NR][MERGE_MAXV];
#pragma unroll
    for (int r = 0; r < BAND_NR; ++r) load_row(dr_[r], wave + r * NW);
    for (int cidx = wave; cidx < n_band; cidx += BAND_NR * NW) {
#pragma unroll
        for (int r = 0; r < BAND_NR; ++r) {
            if (cidx + r * NW < n_band) score_row(dr_[r], cidx + r * NW);
            load_row(dr_[r], cidx + (r + BAND_NR) * NW);
        }
    }
    __syncthreads();
}

// this wave's share of L.keys[0, n_band) folded into `best` (a sorted top-64 held across calls; first: no keys folded yet)
template <int NT>
__device__ __forceinline__ void band_fold64(const BandLds& L, int n_band, uint64_t& best, bool& first) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = wave * 64; base < n_band; base += NW * 64) {
        const uint64_t key = base + lane < n_band ? L.keys[base + lane] : KEY_NONE;
        best = first ? wave_sort_desc(key) : wave_merge_top64(best, key, lane);
        first = false;
    }
}

// the waves' sorted top-64 lists merged by wave 0, the query's k <= 64 results emitted
template <int NT>
__device__ __forceinline__ void band_emit64(const SearchArgs& p, int q, BandLds& L, uint64_t best, bool first) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                                                     // (everyone has read its keys: the head of L.keys is reused)
    L.keys[wave * 64 + lane] = first ? KEY_NONE : best;
    __syncthreads();
    if (wave == 0) {
        uint64_t top = L.keys[lane];
        for (int w = 1; w < NW; ++w) top = wave_merge_top64(top, L.keys[w * 64 + lane], lane);
        if (lane < p.k) emit_slot(p, q, lane, top);
    }
}

// The whole band pass of query q inside its own workgroup, ANY band size (k <= 64): one pass over the score row when the band
// fits BAND_MAX rows, else — or with tau unknown (-inf) — the index in segments of BAND_MAX rows (a segment's band always
// fits), the waves' top-64 lists carried across the segments.  Returns the band size.
template <int NT>
__device__ __forceinline__ int band_pass_in_place(const SearchArgs& p, int q, float tau, const float* __restrict__ row, BandLds& L,
                                                  int* ucnt_s) {
    const int lane = threadIdx.x & 63, n_docs = (int)p.n_docs;
    f32x4 qv[MERGE_MAXV];
    load_query_regs(qv, p.q_f32 + (size_t)q * p.dim, p.dim >> 2, lane);
    uint64_t best = KEY_NONE;
    bool first = true;
    int n_band = band_gather<NT>(row, 0, n_docs, tau, L, ucnt_s);
    if (n_band > BAND_MAX) {
        // One workgroup re-scoring tens of thousands of fp32 rows takes milliseconds (100 000 rows: 70 ms; the exact pass sweeps
        // the fp32 index with the whole chip: ~1 ms per 8 queries).  When the exact pass follows this launch the query goes on
        // its list; when it does not (the common case: its two idle launches were 7 us of a 105 us search) the walk below is
        // done once and the host-visible word makes the engine launch the exact pass behind this index's searches from now on.
        if (p.exact_follows) {
            if (threadIdx.x == 0) {
                const int pos = atomicAdd(p.flag2_count, 1);
                p.flag2_list[pos] = q;
                if (p.stats) atomicAdd(&p.stats[5], 1u);
            }
            return n_band;
        }
        if (threadIdx.x == 0) {
            if (p.huge_seen) __hip_atomic_store(p.huge_seen, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (p.stats) atomicAdd(&p.stats[5], 1u);          // (counted like the ones the exact pass takes)
        }
    }
    if (n_band <= BAND_MAX) {
        band_rescore<NT>(p, qv, n_band, L);
        band_fold64<NT>(L, n_band, best, first);
    } else {
        for (int lo = 0; lo < n_docs; lo += BAND_MAX) {
            const int nb = band_gather<NT>(row, lo, min(n_docs, lo + BAND_MAX), tau, L, ucnt_s);
            band_rescore<NT>(p, qv, nb, L);
            band_fold64<NT>(L, nb, best, first);
        }
    }
    band_emit64<NT>(p, q, L, best, first);
    return n_band;
}

}  // namespace vr
