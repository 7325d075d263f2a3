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
constexpr int MERGE_CAP = 1024;    // survivor buffer per query (merge kernels): what the certification can re-score in place
#ifndef VR_MERGE_GD
#define VR_MERGE_GD 16
#endif
// gather depth of the merge kernels beyond KP: the first gather keeps every list entry above (a lower bound of) the
// (KP + MERGE_GD_EXTRA)-th best key — deep enough that the certification rarely has to gather again
constexpr int MERGE_GD_EXTRA = VR_MERGE_GD;

// ---- exact fp32 scores ------------------------------------------------------------------------
// ONE definition of the fp32 dot product behind every score the library returns: lane l owns the
// float4 chunks l, l + 64, ...; inside a chunk a fixed fma chain; wave_sum over the lanes last.
// (Written with explicit fmaf so that the compiler's contraction choices cannot differ between the
// kernels that use it: the merge kernels' re-scoring and the exact pass of search_exact.hip must
// produce the SAME float for the same (query, row).)
__device__ __forceinline__ float dot_chunk(const f32x4 q, const f32x4 d, float a) {
    float t = q[0] * d[0];
    t = __builtin_fmaf(q[1], d[1], t);
    t = __builtin_fmaf(q[2], d[2], t);
    t = __builtin_fmaf(q[3], d[3], t);
    return a + t;
}
__device__ __forceinline__ void load_query_regs(f32x4 (&qv)[MERGE_MAXV], const float* q, int nv, int lane) {
    const f32x4* qr = reinterpret_cast<const f32x4*>(q);
#pragma unroll
    for (int i = 0; i < MERGE_MAXV; ++i) {
        const int c = lane + i * 64;
        qv[i] = (c < nv) ? qr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
__device__ __forceinline__ float dot_lane(const f32x4 (&qv)[MERGE_MAXV], const float* row, int nv, int lane) {
    const f32x4* dr = reinterpret_cast<const f32x4*>(row);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < MERGE_MAXV; ++i) {
        const int cc = lane + i * 64;
        if (cc < nv) a = dot_chunk(qv[i], dr[cc], a);
    }
    return a;
}

__device__ __forceinline__ float key_score(uint64_t key) { return orderable_f32((uint32_t)(key >> 32)); }

// slot `slot` of query q's result: (score, row id) or, for the multi-GPU exchange, the packed key with the
// shard's id offset applied
__device__ __forceinline__ void emit_slot(const SearchArgs& p, int q, int slot, uint64_t key) {
    const bool ok = key != KEY_NONE;
    const size_t o = (size_t)q * p.k + slot;
    if (p.out_keys) {
        const uint32_t gid = (uint32_t)((int64_t)(~(uint32_t)key) + p.id_offset);
        p.out_keys[o] = ok ? ((key & 0xFFFFFFFF00000000ull) | (uint32_t)(~gid)) : KEY_NONE;
    } else {
        p.out_scores[o] = ok ? key_score(key) : -INFINITY;
        p.out_ids[o] = ok ? (int64_t)(~(uint32_t)key) : (int64_t)-1;
    }
}

// ---- certified candidate selection ------------------------------------------------------------
// The sweeps rank rows by their bf16-MFMA score b(d); the library returns the ranking by the fp32 score
// s(d).  With q^ = bf16(q) = q + dq, d^ = bf16(d) + ... (round to nearest even: |dx_i| <= 2^-8 |x_i| — bf16 keeps
// 8 significand bits, its unit roundoff is 2^-8, NOT 2^-9) and fp32 accumulation of the exact bf16 products,
//     b(d) - s(d) = (fl(q^.d^) - q^.d^) + (dq.d + q.dd + dq.dd) + (q.d - fl(q.d))
//     |b(d) - s(d)| <= |dq| |d| + |q| |dd| + |dq| |dd| + acc_rel (|q| + |dq|) (|d| + |dd|)              (Cauchy-Schwarz)
// with acc_rel = (2 dim + 128) 2^-24 covering both fp32 summations in any order (<= 2^-23 per add allows a truncating
// adder).  The default bound (query_eps, eps_data) uses the MEASURED residual norms: |dq| of the query at hand (computed
// here), max |d| and max |dd| = max |d - bf16(d)| over the index rows (vr_index_add keeps both) — rigorous for the data
// that is there, and ~2.5x tighter than the worst case 2^-7 + 2^-16 (search_default_eps_rel), because a random
// significand sits 0.29 ulp from its rounding on average, not half an ulp.  A caller's own model: eps_rel |q| max|d|
// (vr_index_set_search_eps).  A row outside
// the re-scored set R can only belong to the fp32 top-k if s(d) >= s_k (the k-th best fp32 score inside
// R), i.e. if b(d) >= s_k - eps =: tau.  certify_tail therefore
//   1. re-scores the bf16 top-KP, takes s_k and tau;
//   2. re-scores every further gathered candidate with b >= tau (they are sorted: a prefix);
//   3. certifies the result if everything NOT re-scored is known to lie below tau:
//        coverB  >= b of every list entry that was not gathered into `cand` (if that bound is too high but the
//                 lists are complete down to tau, the entries >= tau are gathered again first),
//        dropB   >= b of every row the sweep dropped before it reached a list;
//      otherwise the query goes on the flag list and the exact fp32 pass (search_exact.hip) redoes it.
// cand: LDS[64], the best gathered keys sorted descending (KEY_NONE padded); exact_s: LDS[64] scratch.
// regather(tau) -> n: gathers EVERY list entry with a score >= tau again (the first gather stopped at a bound
// that turned out to lie above tau), leaves the best 64 of the n sorted in cand; used when the lists themselves
// are complete down to tau (dropB < tau), so that only a query with more than 64 rows inside the error band —
// or incomplete lists — pays for the exact pass.
// Called by all threads of the query's workgroup (4 waves, or 16 for the handful-of-queries merge: the fp32 re-scoring of the
// candidates is spread over its waves); cand / coverB / dropB must be visible (barrier).
// surv / exact_w: LDS[MERGE_CAP] — what regather() gathered (unsorted) and scratch for its exact keys: when more than 64
// rows lie inside the error band (a cluster of near-duplicate pages) but no more than MERGE_CAP, ALL of them are re-scored
// here and the top k taken from that — the exact pass over the whole index is for what exceeds even this.

// bf16 rounding residual of x (exact in fp32: x and bf16(x) agree in their leading bits)
__device__ __forceinline__ float bf16_resid(float x) { return x - bf2f(f2bf(x)); }

// eps of the certification for the query whose float4 chunks the wave holds in qv (all 64 lanes take part);
// slack factors cover the fp32 rounding of the norms themselves
__device__ __forceinline__ float query_eps(const SearchArgs& p, const f32x4 (&qv)[MERGE_MAXV]) {
    float qq = 0.f, dd = 0.f;
#pragma unroll
    for (int i = 0; i < MERGE_MAXV; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = qv[i][r], e = bf16_resid(x);
            qq = __builtin_fmaf(x, x, qq);
            dd = __builtin_fmaf(e, e, dd);
        }
    const float qn = sqrtf(wave_sum(qq)) * 1.00001f;
    if (!p.eps_data) return p.eps_rel * qn * p.dmax[0];
    const float dq = sqrtf(wave_sum(dd)) * 1.00001f;
    const float dn = p.dmax[0], dr = p.dmax[1];
    return (dq * dn + qn * dr + dq * dr + p.acc_rel * (qn + dq) * (dn + dr)) * 1.00001f;
}

// a query that could not be certified goes on the flag list with its tau and its bf16 row (the band pass's GEMM operand);
// called by the whole workgroup with `flagged` workgroup-uniform; sh_pos: LDS scratch
__device__ __forceinline__ void flag_query(const SearchArgs& p, int q, bool flagged, float tau, int* sh_pos) {
    if (!flagged || !p.flag_count) return;
    if (threadIdx.x == 0) {
        const int pos = atomicAdd(p.flag_count, 1);
        p.flag_list[pos] = q;
        if (p.flag_tau) p.flag_tau[pos] = tau;
        *sh_pos = pos;
    }
    __syncthreads();
    if (p.flag_q) {
        const int pos = *sh_pos;
        const uint32_t* src = reinterpret_cast<const uint32_t*>((const char*)p.q_bf16 + (size_t)q * p.dim * 2);
        uint32_t* dst = reinterpret_cast<uint32_t*>((char*)p.flag_q + (size_t)pos * p.dim * 2);
        for (int c = threadIdx.x; c < p.dim / 2; c += blockDim.x) dst[c] = src[c];
    }
}

// descending bitonic sort of n (power of two) keys in LDS by the whole workgroup
__device__ __forceinline__ void block_bitonic_desc(uint64_t* keys, int n, int tid, int nthreads) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n; i += nthreads) {
                const int partner = i ^ j;
                if (partner > i) {
                    const uint64_t a = keys[i], b = keys[partner];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { keys[i] = b; keys[partner] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// Returns whether the query could NOT be certified (workgroup-uniform; *sh_tau holds its tau).  defer_flag: the caller redoes
// such a query itself (the handful-of-queries merge: search_band.h) — it is counted, but not put on the flag list.
template <int KP, typename Regather>
__device__ __forceinline__ bool certify_tail(const SearchArgs& p, int q, const uint64_t* cand, uint64_t* exact_s,
                                             float coverB, float dropB, float* sh_tau, int* sh_x, Regather regather,
                                             const uint64_t* surv, uint64_t* exact_w, bool defer_flag = false, f32x4* q_s = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int nv = p.dim >> 2;
    const float* qrow = p.q_f32 + (size_t)q * p.dim;
#ifndef VR_RESCORE_PAIRS
#define VR_RESCORE_PAIRS 1
#endif
    // q_s (LDS, MERGE_MAXV * 64 float4, the 256-thread merge of the big sweeps): the bf16 top-KP is re-scored with TWO candidate
    // rows per wave in flight — a wave has KP / 4 of these 9 KB rows to fetch, each a memory round trip — and the query read
    // from LDS meanwhile, so that the two rows' 80 registers do not push the kernel past 128 (four workgroups per CU; with the
    // query in registers too: 207).  Up to 512 queries only: 256 queries 55.5 -> 45.5 us per merge, but with 1 000 workgroups
    // resident the memory system is what the merge waits for — twice the requests in flight: 69.5 -> 83 us (tools/r6/merge_pairs.sh)
    const bool pairs = VR_RESCORE_PAIRS && q_s != nullptr && nw * 2 <= KP && gridDim.x <= 512;
    if (pairs) {
        if (tid < 64) exact_s[tid] = KEY_NONE;
        for (int c = tid; c < MERGE_MAXV * 64; c += blockDim.x)
            q_s[c] = c < nv ? reinterpret_cast<const f32x4*>(qrow)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        for (int c = wave; c < KP; c += 2 * nw) {
            const uint64_t k0 = cand[c], k1 = cand[c + nw];
            const uint32_t i0 = k0 != KEY_NONE ? ~(uint32_t)k0 : 0u, i1 = k1 != KEY_NONE ? ~(uint32_t)k1 : 0u;
            const f32x4* r0 = reinterpret_cast<const f32x4*>(p.index_f32 + (size_t)i0 * p.dim);
            const f32x4* r1 = reinterpret_cast<const f32x4*>(p.index_f32 + (size_t)i1 * p.dim);
            f32x4 d0[MERGE_MAXV], d1[MERGE_MAXV];
#pragma unroll
            for (int i = 0; i < MERGE_MAXV; ++i) {
                const int cc = lane + i * 64;
                d0[i] = cc < nv ? r0[cc] : f32x4{0.f, 0.f, 0.f, 0.f};
                d1[i] = cc < nv ? r1[cc] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __builtin_amdgcn_sched_barrier(0);               // (all loads of both rows are out before the first fma waits for one)
            float p0 = 0.f, p1 = 0.f;
#pragma unroll
            for (int i = 0; i < MERGE_MAXV; ++i) {
                const int cc = lane + i * 64;
                if (cc < nv) { const f32x4 qc = q_s[cc]; p0 = dot_chunk(qc, d0[i], p0); p1 = dot_chunk(qc, d1[i], p1); }   // dot_lane's chain
            }
            const float a0 = wave_sum(p0), a1 = wave_sum(p1);
            if (lane == 0) {
                if (k0 != KEY_NONE) exact_s[c] = make_key(a0, i0);
                if (k1 != KEY_NONE) exact_s[c + nw] = make_key(a1, i1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 qv[MERGE_MAXV];
    load_query_regs(qv, qrow, nv, lane);
    auto rescore = [&](int c) {                          // wave-uniform c
        const uint64_t key = cand[c];
        if (key == KEY_NONE) return;
        const uint32_t id = ~(uint32_t)key;
        const float a = wave_sum(dot_lane(qv, p.index_f32 + (size_t)id * p.dim, nv, lane));
        if (lane == 0) exact_s[c] = make_key(a, id);
    };
    auto below = [](float bound, float tau) { return bound == -INFINITY || bound < tau; };
    if (!pairs) {
        if (tid < 64) exact_s[tid] = KEY_NONE;
        __syncthreads();
        for (int c = wave; c < KP; c += nw) rescore(c);
    }
    __syncthreads();
    const bool certify = p.eps_data || p.eps_rel >= 0.f;
    if (wave == 0) {
        float tau = -INFINITY;
        if (certify) {
            const float eps = query_eps(p, qv);
            const uint64_t kth = shfl_u64(wave_sort_desc(exact_s[lane]), p.k - 1);
            if (kth != KEY_NONE) tau = key_score(kth) - eps;
        }
        if (lane == 0) { *sh_tau = tau; *sh_x = (certify && !below(coverB, tau) && below(dropB, tau) && tau > -INFINITY) ? 1 : 0; }
    }
    __syncthreads();
    const float tau = *sh_tau;
    const bool again = *sh_x != 0;
    __syncthreads();
    if (again) {                                         // workgroup-uniform
        if (tid == 0 && p.stats) atomicAdd(&p.stats[4], 1u);
        const int n = regather(tau);
        if (n > 64 && n <= MERGE_CAP && key_score(cand[63]) >= tau) {
            // the band holds more than the 64 sorted candidates, but all n of its rows sit in surv: re-score every one
            for (int c = wave; c < n; c += nw) {
                const uint64_t key = surv[c];
                const uint32_t id = ~(uint32_t)key;
                const float a = wave_sum(dot_lane(qv, p.index_f32 + (size_t)id * p.dim, nv, lane));
                if (lane == 0) exact_w[c] = make_key(a, id);
            }
            __syncthreads();
            if (wave == 0) {
                uint64_t best = KEY_NONE;
                for (int base = 0; base < n; base += 64) {
                    const uint64_t key = base + lane < n ? exact_w[base + lane] : KEY_NONE;
                    best = base == 0 ? wave_sort_desc(key) : wave_merge_top64(best, key, lane);
                }
                if (lane < p.k) emit_slot(p, q, lane, best);
                if (lane == 0 && p.stats) atomicAdd(&p.stats[1], 1u);
            }
            return false;
        }
        coverB = n > 64 ? key_score(cand[63]) : -INFINITY;
    }
    if (wave == 0) {
        const uint64_t c = cand[lane];
        const int x = certify ? __popcll(__ballot(lane >= KP && c != KEY_NONE && key_score(c) >= tau)) : 0;
        if (lane == 0) *sh_x = x;
    }
    __syncthreads();
    const int x = *sh_x;
    for (int c = KP + wave; c < KP + x; c += nw) rescore(c);
    __syncthreads();
    if (wave == 0) {
        const uint64_t ex = wave_sort_desc(exact_s[lane]);
        if (lane < p.k) emit_slot(p, q, lane, ex);
        if (lane == 0) {
            int what = 3;
            if (certify) what = (below(coverB, tau) && below(dropB, tau)) ? ((x || again) ? 1 : 0) : 2;
            if (p.stats) atomicAdd(&p.stats[what], 1u);
        }
    }
    // (coverB / dropB / tau / x are workgroup-uniform: every thread evaluates the same predicate)
    const bool flagged = certify && !(below(coverB, tau) && below(dropB, tau));
    if (!defer_flag) flag_query(p, q, flagged, tau, sh_x);
    return flagged;
}

}  // namespace vr
