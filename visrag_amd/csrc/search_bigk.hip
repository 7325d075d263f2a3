// vr_index_search for k > 26 (the reference's --retrieve_depth is free: eval.sh uses 10, TREC runs
// are commonly 100 or 1000 deep) and the matching multi-GPU merge for k > 64.
//
// The fused sweeps keep one wave-wide candidate list per (query, chunk), which bounds k at 26.
// Deep retrieval is not the throughput case, so it takes the plain route, exact all the same:
//   1. scores S[q][doc] of a block of <= 256 queries against the whole index with the bf16 MFMA
//      GEMM (EPI_F32), the same arithmetic the sweeps use;
//   2. per query (one workgroup): 3-pass radix select (11 + 11 + 10 bits of the order-preserving
//      key, histograms in LDS) of the K'-th largest score, K' = k + 24 — the margin that keeps a
//      true top-k row inside the candidate set although S carries bf16 rounding;
//   3. gather the K' candidates in row order (all keys above the K'-th, then the lowest ids among
//      its ties), re-score them against the fp32 index (exact fp32 dots, same summation order as
//      rescore_emit), bitonic-sort the keys (score desc, id asc) in LDS, emit the top k.
#include <algorithm>

#include "kernels.h"
#include "search_common.h"

namespace vr {

constexpr int BIGK_CAND = 1024;             // K' = k + margin <= BIGK_CAND
constexpr int BIGK_MARGIN = 24;

int search_bigk_max() { return BIGK_CAND - BIGK_MARGIN; }

// descending bitonic sort of n (power of two, <= 8192) keys in LDS by the whole workgroup
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

__global__ __launch_bounds__(256) void bigk_select_kernel(const float* __restrict__ S, size_t ldS, int n_docs, int kp,
                                                          int k, const float* __restrict__ index_f32,
                                                          const float* __restrict__ q_f32, int dim,
                                                          float* __restrict__ out_scores,
                                                          int64_t* __restrict__ out_ids) {
    __shared__ unsigned hist[2048];
    __shared__ int cand[BIGK_CAND];
    __shared__ uint64_t keys[BIGK_CAND];
    __shared__ unsigned sh_prefix, sh_mask;
    __shared__ int sh_rank, wc[4][2], run[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x;
    const float* row = S + (size_t)q * ldS;

    // ---- 1. radix select: key of the kp-th largest score
    if (tid == 0) { sh_prefix = 0u; sh_mask = 0u; sh_rank = kp; }
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass == 0 ? 21 : pass == 1 ? 10 : 0;
        const int nb = pass < 2 ? 2048 : 1024;
        for (int i = tid; i < 2048; i += 256) hist[i] = 0u;
        __syncthreads();
        const unsigned prefix = sh_prefix, mask = sh_mask;
        for (int i = tid; i < n_docs; i += 256) {
            const unsigned key = f32_orderable(row[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & (nb - 1)], 1u);
        }
        __syncthreads();
        if (wave == 0) {
            // bins from the top: lane l owns bins [hi - 32*(l+1) + ..]: do it in strides of 64 bins
            int rank = sh_rank, sel = -1;
            for (int b0 = nb - 64; b0 >= 0 && sel < 0; b0 -= 64) {
                const unsigned h = hist[b0 + 63 - lane];            // lane 0 = highest bin of the stride
                unsigned incl = h;                                   // inclusive prefix over lanes (from the top)
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
                const unsigned total = __shfl(incl, 63, 64);
                if ((int)total >= rank) {
                    const unsigned long long hit = __ballot((int)incl >= rank);
                    const int l = __ffsll((long long)hit) - 1;
                    const unsigned before = __shfl(incl, l, 64) - __shfl(h, l, 64);
                    sel = b0 + 63 - l;
                    rank -= (int)before;
                } else {
                    rank -= (int)total;
                }
            }
            if (lane == 0) {
                sh_prefix = prefix | ((unsigned)sel << shift);
                sh_mask = mask | ((unsigned)(nb - 1) << shift);
                sh_rank = rank;
            }
        }
        __syncthreads();
    }
    const unsigned T = sh_prefix;               // key of the kp-th largest
    const int need_eq = sh_rank;                // how many keys == T belong to the kp best (lowest ids first)
    const int G = kp - need_eq;                 // keys > T
    // ---- 2. gather in row order
    if (tid < 2) run[tid] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n_docs; i0 += 256) {
        const int i = i0 + tid;
        const unsigned key = i < n_docs ? f32_orderable(row[i]) : 0u;
        const bool gt = i < n_docs && key > T, eq = i < n_docs && key == T;
        const unsigned long long bg = __ballot(gt), be = __ballot(eq);
        if (!__syncthreads_or(gt || eq)) continue;                   // (barrier; most blocks hold no candidate)
        if (lane == 0) { wc[wave][0] = __popcll(bg); wc[wave][1] = __popcll(be); }
        __syncthreads();
        int og = run[0], oe = run[1], tg = 0, te = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) { og += wc[w][0]; oe += wc[w][1]; }
            tg += wc[w][0]; te += wc[w][1];
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        if (gt) cand[og + __popcll(bg & below)] = i;
        if (eq) { const int e = oe + __popcll(be & below); if (e < need_eq) cand[G + e] = i; }
        __syncthreads();
        if (tid == 0) { run[0] += tg; run[1] += te; }
        __syncthreads();
    }
    // ---- 3. exact fp32 re-scoring, sort, emit
    int n2 = 1;
    while (n2 < kp) n2 <<= 1;
    for (int c = kp + tid; c < n2; c += 256) keys[c] = KEY_NONE;
    const int nv = dim >> 2;
    const f32x4* qr = reinterpret_cast<const f32x4*>(q_f32 + (size_t)q * dim);
    f32x4 qv[MERGE_MAXV];
#pragma unroll
    for (int i = 0; i < MERGE_MAXV; ++i) {
        const int c = lane + i * 64;
        qv[i] = (c < nv) ? qr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int c = wave; c < kp; c += 4) {
        const int id = cand[c];
        const f32x4* dr = reinterpret_cast<const f32x4*>(index_f32 + (size_t)id * dim);
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < MERGE_MAXV; ++i) {
            const int cc = lane + i * 64;
            if (cc < nv) {
                const f32x4 d = dr[cc];
                a += qv[i][0] * d[0] + qv[i][1] * d[1] + qv[i][2] * d[2] + qv[i][3] * d[3];
            }
        }
        a = wave_sum(a);
        if (lane == 0) keys[c] = make_key(a, (uint32_t)id);
    }
    __syncthreads();
    block_bitonic_desc(keys, n2, tid, 256);
    for (int c = tid; c < k; c += 256) {
        const uint64_t key = c < kp ? keys[c] : KEY_NONE;
        const bool ok = key != KEY_NONE;
        out_scores[(size_t)q * k + c] = ok ? orderable_f32((uint32_t)(key >> 32)) : -INFINITY;
        out_ids[(size_t)q * k + c] = ok ? (int64_t)(~(uint32_t)key) : (int64_t)-1;
    }
}

// queries [q0, q0 + nq_block) of the call: S holds their bf16-MFMA scores, one row of ldS floats each
hipError_t launch_search_bigk(const SearchArgs& a, const float* S, size_t ldS, int q0, int nq_block, hipStream_t s) {
    if (nq_block <= 0) return hipSuccess;
    if (a.k > search_bigk_max() || a.dim % 4 || a.dim > 64 * 4 * MERGE_MAXV) return hipErrorInvalidValue;
    const int kp = (int)std::min<int64_t>(a.n_docs, a.k + BIGK_MARGIN);
    hipLaunchKernelGGL(bigk_select_kernel, dim3(nq_block), dim3(256), 0, s, S, ldS, (int)a.n_docs, kp, a.k, a.index_f32,
                       a.q_f32 + (size_t)q0 * a.dim, a.dim, a.out_scores + (size_t)q0 * a.k, a.out_ids + (size_t)q0 * a.k);
    return hipGetLastError();
}

// ---- multi-GPU merge for k > 64: one workgroup per query, all n_parts * k keys sorted in LDS --------
constexpr int MERGE_BIG_MAX = 8192;

__global__ __launch_bounds__(256) void topk_merge_big_kernel(const float* __restrict__ scores,
                                                             const int64_t* __restrict__ ids, int n_parts, int nq,
                                                             int k, int n2, float* __restrict__ out_scores,
                                                             int64_t* __restrict__ out_ids) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw);
    const int q = blockIdx.x, tid = threadIdx.x;
    const int total = n_parts * k;
    for (int e = tid; e < n2; e += 256) {
        uint64_t key = KEY_NONE;
        if (e < total) {
            const int part = e / k, sidx = e % k;
            const size_t o = ((size_t)part * nq + q) * k + sidx;
            if (ids[o] >= 0) key = make_key(scores[o], (uint32_t)ids[o]);
        }
        keys[e] = key;
    }
    __syncthreads();
    block_bitonic_desc(keys, n2, tid, 256);
    for (int c = tid; c < k; c += 256) {
        const uint64_t key = keys[c];
        const bool ok = key != KEY_NONE;
        out_scores[(size_t)q * k + c] = ok ? orderable_f32((uint32_t)(key >> 32)) : -INFINITY;
        out_ids[(size_t)q * k + c] = ok ? (int64_t)(~(uint32_t)key) : (int64_t)-1;
    }
}

hipError_t launch_topk_merge_big(const float* scores, const int64_t* ids, int n_parts, int nq, int k, float* out_scores,
                                 int64_t* out_ids, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    const long total = (long)n_parts * k;
    if (k <= 0 || total > MERGE_BIG_MAX) return hipErrorInvalidValue;
    int n2 = 64;
    while (n2 < total) n2 <<= 1;
    static unsigned long long attr = 0;     // bit d: set on device d
    set_max_dynamic_lds((const void*)topk_merge_big_kernel, MERGE_BIG_MAX * 8, attr);
    hipLaunchKernelGGL(topk_merge_big_kernel, dim3(nq), dim3(256), (size_t)n2 * 8, s, scores, ids, n_parts, nq, k, n2,
                       out_scores, out_ids);
    return hipGetLastError();
}

}  // namespace vr
