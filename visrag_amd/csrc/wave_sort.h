// 64-lane bitonic sort of 64-bit keys without LDS round trips.
//
// The compare-exchange partner of lane l at distance j is lane l ^ j.  On gfx950 every such
// exchange has a register-to-register form:
//   j = 1, 2   DPP quad_perm
//   j = 4      DPP row_half_mirror (i -> 7 - i) followed by quad_perm [3,2,1,0]   (= i ^ 4)
//   j = 8      DPP row_ror:8 (rotation by half a 16-lane row = i ^ 8)
//   j = 16     v_permlane16_swap_b32 (swaps odd rows of one register with even rows of the other)
//   j = 32     v_permlane32_swap_b32 (swaps the upper half-wave of one register with the lower
//              half-wave of the other)
// against two ds_bpermute_b32 (LDS crossbar, ~100+ cycles each with one wave per SIMD) per stage
// for __shfl_xor.  21 stages sort 64 keys; one key per lane, lane 0 ends with the largest.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vr {

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}

// partner value of a DPP-reachable distance (1, 2, 4, 8)
template <int J>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t v) {
    static_assert(J == 1 || J == 2 || J == 4 || J == 8, "DPP distances only");
    if constexpr (J == 1) return dpp_mov<0xB1>(v);                       // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return dpp_mov<0x4E>(v);                  // quad_perm [2,3,0,1]
    else if constexpr (J == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(v));  // half_mirror, then [3,2,1,0]
    else return dpp_mov<0x128>(v);                                       // row_ror:8
}

// lanes that keep the LARGER key of their pair in stage (K, J) of a descending sort
template <int K, int J>
constexpr uint64_t keep_max_mask() {
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l)
        if (((l & K) == 0) == ((l & J) == 0)) m |= 1ull << l;
    return m;
}

// One compare-exchange stage.  Every form yields a pair (x, y) = {own, partner} in some order;
// the lane keeps  max(x, y)  if it is a keep-max lane, else  min(x, y):
//   keep x  <=>  (x > y) XOR (lane is keep-min)        (x == y: either is right)
// i.e. one 64-bit compare, one scalar XOR with a compile-time lane mask, two v_cndmask.
template <int K, int J>
__device__ __forceinline__ uint64_t bitonic_stage(uint64_t key) {
    uint64_t x, y;
    if constexpr (J <= 8) {
        x = key;
        y = ((uint64_t)lane_xor_u32<J>((uint32_t)(key >> 32)) << 32) | lane_xor_u32<J>((uint32_t)key);
    } else if constexpr (J == 16) {
        const auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)key, (uint32_t)key, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((uint32_t)(key >> 32), (uint32_t)(key >> 32), false, false);
        x = ((uint64_t)hi[0] << 32) | lo[0];
        y = ((uint64_t)hi[1] << 32) | lo[1];
    } else {
        const auto lo = __builtin_amdgcn_permlane32_swap((uint32_t)key, (uint32_t)key, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((uint32_t)(key >> 32), (uint32_t)(key >> 32), false, false);
        x = ((uint64_t)hi[0] << 32) | lo[0];
        y = ((uint64_t)hi[1] << 32) | lo[1];
    }
    const uint64_t m = __builtin_amdgcn_ballot_w64(x > y) ^ ~keep_max_mask<K, J>();
    return __builtin_amdgcn_inverse_ballot_w64(m) ? x : y;
}

template <int K, int J>
__device__ __forceinline__ uint64_t bitonic_merge_steps(uint64_t key) {
    key = bitonic_stage<K, J>(key);
    if constexpr (J > 1) key = bitonic_merge_steps<K, J / 2>(key);
    return key;
}

template <int K>
__device__ __forceinline__ uint64_t bitonic_levels(uint64_t key) {
    if constexpr (K > 2) key = bitonic_levels<K / 2>(key);
    return bitonic_merge_steps<K, K / 2>(key);
}

// full sort, descending
__device__ __forceinline__ uint64_t wave_sort_desc(uint64_t key) { return bitonic_levels<64>(key); }

// `key` holds a bitonic sequence over the 64 lanes -> sorted descending (6 stages)
__device__ __forceinline__ uint64_t wave_bitonic_finish_desc(uint64_t key) {
    return bitonic_merge_steps<64, 32>(key);
}

}  // namespace vr
