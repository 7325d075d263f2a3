// Small arithmetic of the generator's decode step that more than one kernel performs (the separate launches of gen.hip and
// the persistent kernel of gen_persist.hip must produce the same bits): spelled out operation by operation, because left to
// -ffp-contract=fast hipcc fuses `a * b + c * d` one way in one kernel and the other way in the next (seen: a rotated q / k
// element one ulp apart, a bf16 rounding flipped, logits 2e-3 of their scale apart at one decode step in a few hundred).
#pragma once
#include "common.h"

namespace vr {

// rotate-half pair (x1, x2) by the angle whose cosine / sine are cs / sn: x * cos + rotate_half(x) * sin
__device__ __forceinline__ void rope_rotate(float& x1, float& x2, float cs, float sn) {
#pragma clang fp contract(off)
    const float a = x2 * sn, b = x1 * sn;
    const float r1 = __builtin_fmaf(x1, cs, -a), r2 = __builtin_fmaf(x2, cs, b);
    x1 = r1; x2 = r2;
}

// sum of the squares of a float4 (one lane's share of a row's sum of squares)
__device__ __forceinline__ float sumsq4(f32x4 v) {
#pragma clang fp contract(off)
    return __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[1], v[1], v[0] * v[0])));
}

// one KV range's contribution to the merge of partial attention rows: weight e = 2^(lse - max)
__device__ __forceinline__ void merge_range(float& num, float& den, float e, float pv) {
#pragma clang fp contract(off)
    num = __builtin_fmaf(e, pv, num);
    den = den + e;
}

}  // namespace vr
