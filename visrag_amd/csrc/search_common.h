// Shared device helpers of the fused search kernels (search.hip, search256.hip): orderable score
// keys and the 64-lane bitonic network.
#pragma once
#include "common.h"
#include "kernels.h"
#include "wave_sort.h"

namespace vr {

constexpr int SRCH_CAP = 64;        // per-query LDS candidate buffer (one entry per lane)
constexpr int SRCH_TRIG = 32;       // compact when a buffer holds more than this

__device__ __forceinline__ uint32_t f32_orderable(float f) {
    const uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float orderable_f32(uint32_t o) {
    const uint32_t u = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(u);
}
// sort key: larger key = better candidate (higher score, then LOWER id)
__device__ __forceinline__ uint64_t make_key(float score, uint32_t id) {
    return ((uint64_t)f32_orderable(score) << 32) | (uint32_t)(~id);
}
constexpr uint64_t KEY_NONE = 0;    // below every real key (score -inf, id 0xffffffff -> ~ = 0 ...)

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m) {
    const uint32_t lo = __shfl_xor((uint32_t)v, m, 64);
    const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), m, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src) {
    const uint32_t lo = __shfl((uint32_t)v, src, 64);
    const uint32_t hi = __shfl((uint32_t)(v >> 32), src, 64);
    return ((uint64_t)hi << 32) | lo;
}

// 64-lane bitonic sort, descending: lane 0 ends with the largest key (wave_sort.h: DPP /
// permlane-swap exchanges, no LDS round trips).
__device__ __forceinline__ uint64_t wave_bitonic_desc(uint64_t key, int /*lane*/) { return wave_sort_desc(key); }
// top-64 of (sorted-desc `cur`) U (arbitrary `fresh`), sorted descending: sort `fresh` ASCENDING
// (descending sort of the complemented keys); max(cur[i], fresh_asc[i]) is then a bitonic
// sequence holding the 64 largest keys, which the last 6 stages of the network order.
__device__ __forceinline__ uint64_t wave_merge_top64(uint64_t cur, uint64_t fresh, int /*lane*/) {
    const uint64_t asc = ~wave_sort_desc(~fresh);
    return wave_bitonic_finish_desc(cur > asc ? cur : asc);
}

constexpr int MERGE_MAXV = 10;     // dim <= 64 * 4 * MERGE_MAXV
constexpr int MERGE_CAP = 256;     // survivor buffer per query (merge kernels)

// Tail of both merge kernels: `best` holds the bf16-score top-KP keys of query q, sorted
// descending (lane c = candidate c).  Re-score them against the fp32 index with exact fp32 dot
// products (the ranking torch.topk over an fp32 matmul sees), sort on (score desc, id asc), emit
// the top k.  Four candidates per round so that their row reads overlap (the rows are cold:
// latency-, not bandwidth-bound).
template <int KP>
__device__ __forceinline__ void rescore_emit(const SearchArgs& p, int q, uint64_t best, int lane) {
    const int nv = p.dim >> 2;
    f32x4 qv[MERGE_MAXV];
    const f32x4* qr = reinterpret_cast<const f32x4*>(p.q_f32 + (size_t)q * p.dim);
#pragma unroll
    for (int i = 0; i < MERGE_MAXV; ++i) {
        const int c = lane + i * 64;
        qv[i] = (c < nv) ? qr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    uint64_t exact = KEY_NONE;
    for (int c0 = 0; c0 < KP; c0 += 4) {
        if (shfl_u64(best, c0) == KEY_NONE) break;       // wave-uniform; keys are sorted, NONE last
        float s[4];
        uint32_t id[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t key = shfl_u64(best, c0 + u);
            ok[u] = key != KEY_NONE;
            id[u] = ok[u] ? ~(uint32_t)key : 0u;        // row 0 is always readable
            const f32x4* dr = reinterpret_cast<const f32x4*>(p.index_f32 + (size_t)id[u] * p.dim);
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < MERGE_MAXV; ++i) {
                const int cc = lane + i * 64;
                if (cc < nv) {
                    const f32x4 d = dr[cc];
                    a += qv[i][0] * d[0] + qv[i][1] * d[1] + qv[i][2] * d[2] + qv[i][3] * d[3];
                }
            }
            s[u] = a;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float t = wave_sum(s[u]);
            if (ok[u] && lane == c0 + u) exact = make_key(t, id[u]);
        }
    }
    exact = wave_sort_desc(exact);
    if (lane < p.k) {
        const bool ok = exact != KEY_NONE;
        p.out_scores[(size_t)q * p.k + lane] = ok ? orderable_f32((uint32_t)(exact >> 32)) : -INFINITY;
        p.out_ids[(size_t)q * p.k + lane] = ok ? (int64_t)(~(uint32_t)exact) : (int64_t)-1;
    }
}

}  // namespace vr
