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
//      its ties), re-score them against the fp32 index (exact fp32 dots: dot_lane), bitonic-sort
//      the keys (score desc, id asc) in LDS;
//   4. certify as the fused path does (search_common.h): if the K'-th bf16 score is not below
//      tau = s_k - eps, gather and re-score EVERY row with a bf16 score >= tau (<= 1024), else
//      flag the query for the exact fp32 pass; emit the top k.
#include <algorithm>

#include "kernels.h"
#include "search_common.h"

namespace vr {

constexpr int BIGK_CAND = 1024;             // K' = k + margin <= BIGK_CAND
constexpr int BIGK_MARGIN = 24;

int search_bigk_max() { return BIGK_CAND - BIGK_MARGIN; }

// exact == 0: S rows are bf16-MFMA scores of queries blockIdx.x; the result is certified like the fused path's
//   (search_common.h): tau = s_k - eps; if the radix threshold T does not lie below tau, every row with a bf16 score
//   >= tau is gathered and re-scored instead (up to BIGK_CAND of them), else the query is flagged.
// exact == 1: S row i holds EXACT fp32 scores of flagged query flag_list[i] (search_exact.hip); plain top-k of it.
__global__ __launch_bounds__(256) void bigk_select_kernel(SearchArgs p, const float* __restrict__ S, size_t ldS, int kp_want,
                                                          int exact, int sub, int max_slots) {
    __shared__ unsigned hist[2048];
    __shared__ int cand[BIGK_CAND];
    __shared__ uint64_t keys[BIGK_CAND];
    __shared__ unsigned sh_prefix, sh_mask;
    __shared__ int sh_rank, wc[4][2], run[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // exact: a fixed grid walks entries [sub, sub + max_slots) of the flag list
    const int n_slots = exact ? min(max(p.flag_count[0] - sub, 0), max_slots) : (int)gridDim.x;
    for (int slot = blockIdx.x; slot < n_slots; slot += gridDim.x) {
    const int q = exact ? p.flag_list[sub + slot] : slot;
    const float* row = S + (size_t)slot * ldS;
    const int n_docs = (int)p.n_docs, k = p.k, dim = p.dim;
    int kp = min(n_docs, kp_want);
    __syncthreads();                                                    // (LDS of the previous slot is free)

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
    // ---- 2. gather in row order: rows with key > hi_T, then the first `eq_take` rows with key == hi_T
    auto gather = [&](unsigned hi_T, int n_gt, int eq_take, int cap) -> int {
        if (tid < 2) run[tid] = 0;
        __syncthreads();
        for (int i0 = 0; i0 < n_docs; i0 += 256) {
            const int i = i0 + tid;
            const unsigned key = i < n_docs ? f32_orderable(row[i]) : 0u;
            const bool gt = i < n_docs && key > hi_T, eq = i < n_docs && key == hi_T && eq_take > 0;
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
            if (gt) { const int g = og + __popcll(bg & below); if (g < cap) cand[g] = i; }
            if (eq) { const int e = oe + __popcll(be & below); if (e < eq_take && n_gt + e < cap) cand[n_gt + e] = i; }
            __syncthreads();
            if (tid == 0) { run[0] += tg; run[1] += te; }
            __syncthreads();
        }
        return run[0];                             // rows with key > hi_T
    };
    gather(T, G, need_eq, BIGK_CAND);
    // ---- 3. exact fp32 re-scoring, sort
    const int nv = dim >> 2;
    f32x4 qv[MERGE_MAXV];
    load_query_regs(qv, p.q_f32 + (size_t)q * dim, nv, lane);
    auto rescore_sort = [&](int m) {
        int n2 = 1;
        while (n2 < m) n2 <<= 1;
        for (int c = m + tid; c < n2; c += 256) keys[c] = KEY_NONE;
        for (int c = wave; c < m; c += 4) {
            const int id = cand[c];
            const float a = wave_sum(dot_lane(qv, p.index_f32 + (size_t)id * dim, nv, lane));
            if (lane == 0) keys[c] = make_key(a, (uint32_t)id);
        }
        __syncthreads();
        block_bitonic_desc(keys, n2, tid, 256);
    };
    rescore_sort(kp);
    // ---- 4. certification (rows outside the re-scored set have a bf16 score <= T's)
    if (!exact && (p.eps_data || p.eps_rel >= 0.f)) {
        int what = 0;
        float tau = -INFINITY;
        if (kp < n_docs) {
            const float eps = query_eps(p, qv);
            tau = key_score(keys[k - 1]) - eps;                       // kp >= k here (kp < n_docs => kp = k + margin)
            if (!(orderable_f32(T) < tau)) {
                // every row whose bf16 score is >= tau (strictly above the key just below tau's)
                __syncthreads();
                const unsigned tk = f32_orderable(tau);
                const int m = gather(tk ? tk - 1u : 0u, 0, 0, BIGK_CAND);
                __syncthreads();
                if (m <= BIGK_CAND && m >= kp) { rescore_sort(m); kp = m; what = 1; }
                else what = 2;
            }
        }
        if (tid == 0 && p.stats) atomicAdd(&p.stats[what], 1u);
        __syncthreads();
        flag_query(p, q, what == 2, tau, &sh_rank);
    } else if (!exact && tid == 0 && p.stats) {
        atomicAdd(&p.stats[3], 1u);
    }
    for (int c = tid; c < k; c += 256) emit_slot(p, q, c, c < kp ? keys[c] : KEY_NONE);
    }
}

// queries [q0, q0 + nq_block) of the call: S holds their bf16-MFMA scores, one row of ldS floats each.
// `a` is the BLOCK's view (q_f32 / outputs / flag lists start at the block's first query).
hipError_t launch_search_bigk(const SearchArgs& a, const float* S, size_t ldS, int q0, int nq_block, hipStream_t s) {
    (void)q0;
    if (nq_block <= 0) return hipSuccess;
    if (a.k > search_bigk_max() || a.dim % 4 || a.dim > 64 * 4 * MERGE_MAXV) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bigk_select_kernel, dim3(nq_block), dim3(256), 0, s, a, S, ldS, a.k + BIGK_MARGIN, 0, 0, 0);
    return hipGetLastError();
}

// exact top-k of the flagged queries from their exact score rows (slot i of S = query flag_list[i])
hipError_t launch_exact_select(const SearchArgs& a, const float* S, size_t ldS, int sub, int max_slots, hipStream_t s) {
    if (max_slots <= 0) return hipSuccess;
    if (a.k > BIGK_CAND || a.dim % 4 || a.dim > 64 * 4 * MERGE_MAXV || !a.flag_count || !a.flag_list) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bigk_select_kernel, dim3(max_slots < 128 ? max_slots : 128), dim3(256), 0, s, a, S, ldS, a.k, 1, sub, max_slots);
    return hipGetLastError();
}

// ---- multi-GPU merge for k > 64: one workgroup per query, all n_parts * k keys sorted in LDS --------
constexpr int MERGE_BIG_MAX = 8192;

__global__ __launch_bounds__(256) void topk_merge_big_kernel(const float* __restrict__ scores,
                                                             const int64_t* __restrict__ ids,
                                                             const unsigned long long* __restrict__ pk, int n_parts,
                                                             int nq, int k, int n2, float* __restrict__ out_scores,
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
            if (pk) key = pk[o];
            else if (ids[o] >= 0) key = make_key(scores[o], (uint32_t)ids[o]);
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

hipError_t launch_topk_merge_any(const float* scores, const int64_t* ids, const unsigned long long* pk, int n_parts, int nq,
                                 int k, float* out_scores, int64_t* out_ids, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    const long total = (long)n_parts * k;
    if (k <= 0 || total > MERGE_BIG_MAX) return hipErrorInvalidValue;
    int n2 = 64;
    while (n2 < total) n2 <<= 1;
    static unsigned long long attr = 0;     // bit d: set on device d
    set_max_dynamic_lds((const void*)topk_merge_big_kernel, MERGE_BIG_MAX * 8, attr);
    hipLaunchKernelGGL(topk_merge_big_kernel, dim3(nq), dim3(256), (size_t)n2 * 8, s, scores, ids, pk, n_parts, nq, k, n2,
                       out_scores, out_ids);
    return hipGetLastError();
}

hipError_t launch_topk_merge_big(const float* scores, const int64_t* ids, int n_parts, int nq, int k, float* out_scores,
                                 int64_t* out_ids, hipStream_t s) {
    return launch_topk_merge_any(scores, ids, nullptr, n_parts, nq, k, out_scores, out_ids, s);
}

}  // namespace vr
