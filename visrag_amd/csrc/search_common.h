// Shared device helpers of the fused search kernels (search.hip, search256.hip): orderable score
// keys and the 64-lane bitonic network.
#pragma once
#include "common.h"
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

}  // namespace vr
