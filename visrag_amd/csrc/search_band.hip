// The band pass: what happens to a query whose candidate lists could not PROVE its fp32 top-k
// (search_common.h: certify_tail puts it on the flag list with its tau = s_k - eps and its bf16 row).
//
// Such a query needs every index row whose bf16-MFMA score is >= tau re-scored in fp32 — nothing else can belong
// to the fp32 top-k — and the fused sweeps cannot deliver that set when it is large (a cluster of near-duplicate
// pages: templated decks, forms) or when their lists dropped rows before tau was known (a pre-pass threshold
// inside the error band, a compacted half-list).  Rounds 1-3 sent these queries through an exact fp32 pass over the
// WHOLE index, one sweep of the fp32 rows per 8 queries (0.15-0.2 ms each at 100k rows).  Now:
//
//   1. S[slot][row] = bf16-MFMA score of flagged query `slot` against every row: ONE bf16 GEMM over the compacted
//      flagged queries (the 8-wave 256^2 kernel with a device-side row count: tiles past the flag count leave at
//      once, so the launch costs a dispatch when nothing is flagged) — the error bound holds for ANY fp32
//      accumulation order of the bf16 products, so these scores need not equal the sweep's bit for bit;
//   2. band_select_kernel (here), one workgroup per flagged query: count the rows with S >= tau; gather their ids;
//      re-score each in fp32 (dot_lane: the one definition of the library's fp32 score, two rows in flight per
//      wave); sort the keys (score desc, id asc) in LDS; emit the top k.  Cost ~ band size x dim x 4 B of row reads.
//   3. a band of more than BAND_MAX rows (8192: half of a 100k-row index scoring the same to 3e-3 is not a
//      retrieval problem any more) goes on a second list, and the exact fp32 pass (search_exact.hip) redoes it.
//
// Roofline: the GEMM is MFMA-bound (2 x flagged x rows x dim), the selection HBM / L2 (score row + band rows).
#include "kernels.h"
#include "search_common.h"

namespace vr {

constexpr int BAND_MAX = 8192;

int search_band_max() { return BAND_MAX; }

__global__ __launch_bounds__(256) void band_select_kernel(SearchArgs p, const float* __restrict__ S, size_t ldS, int sub,
                                                          int max_slots) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw);              // [BAND_MAX]: row ids first, then their exact keys
    __shared__ int cnt_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_slots = min(max(p.flag_count[0] - sub, 0), max_slots);
    const int n_docs = (int)p.n_docs, k = p.k, dim = p.dim, nv = dim >> 2;
    for (int slot = blockIdx.x; slot < n_slots; slot += gridDim.x) {
        const int q = p.flag_list[sub + slot];
        const float tau = p.flag_tau[sub + slot];
        const float* row = S + (size_t)slot * ldS;
        __syncthreads();                                                 // (LDS of the previous slot is free)
        if (tid == 0) cnt_s = 0;
        __syncthreads();
        // ---- 1. size of the band
        int c = 0;
        for (int i = tid; i < n_docs; i += 256) c += row[i] >= tau ? 1 : 0;
        if (c) atomicAdd(&cnt_s, c);
        __syncthreads();
        const int n_band = cnt_s;
        if (n_band > BAND_MAX || !(tau > -INFINITY)) {                   // workgroup-uniform
            if (tid == 0) {
                const int pos = atomicAdd(p.flag2_count, 1);
                p.flag2_list[pos] = q;
                if (p.stats) atomicAdd(&p.stats[5], 1u);
            }
            continue;
        }
        // ---- 2. the band's row ids (any order: the keys are sorted afterwards)
        __syncthreads();
        if (tid == 0) cnt_s = 0;
        __syncthreads();
        for (int i0 = 0; i0 < n_docs; i0 += 256) {
            const int i = i0 + tid;
            const bool in = i < n_docs && row[i] >= tau;
            const unsigned long long b = __ballot(in);
            if (b == 0ull) continue;                                     // wave-uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(&cnt_s, __popcll(b));
            base = __shfl(base, 0, 64);
            if (in) keys[base + __popcll(b & ((1ull << lane) - 1ull))] = (uint64_t)(uint32_t)i;
        }
        __syncthreads();
        // ---- 3. exact fp32 scores: a wave takes rows wave, wave + 4, ...; the next row's loads are in flight while
        //         the current one is reduced
        f32x4 qv[MERGE_MAXV];
        load_query_regs(qv, p.q_f32 + (size_t)q * dim, nv, lane);
        auto load_row = [&](f32x4 (&dv)[MERGE_MAXV], int cidx) {
            if (cidx >= n_band) return;
            const f32x4* dr = reinterpret_cast<const f32x4*>(p.index_f32 + (size_t)(uint32_t)keys[cidx] * dim);
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
                if (cc < nv) a = dot_chunk(qv[i], dv[i], a);              // the chain of dot_lane(q, row)
            }
            a = wave_sum(a);
            if (lane == 0) keys[cidx] = make_key(a, (uint32_t)keys[cidx]);
        };
        f32x4 dA[MERGE_MAXV], dB[MERGE_MAXV];
        load_row(dA, wave);
        for (int cidx = wave; cidx < n_band; cidx += 8) {
            load_row(dB, cidx + 4);
            score_row(dA, cidx);
            if (cidx + 4 >= n_band) break;
            load_row(dA, cidx + 8);
            score_row(dB, cidx + 4);
        }
        int n2 = 64;
        while (n2 < n_band) n2 <<= 1;
        for (int e = n_band + tid; e < n2; e += 256) keys[e] = KEY_NONE;
        __syncthreads();
        // ---- 4. sort, emit
        block_bitonic_desc(keys, n2, tid, 256);
        for (int e = tid; e < k; e += 256) emit_slot(p, q, e, e < n_band ? keys[e] : KEY_NONE);
    }
}

hipError_t launch_band_select(const SearchArgs& a, const float* S, size_t ldS, int sub, int max_slots, hipStream_t s) {
    if (max_slots <= 0 || a.n_docs <= 0) return hipSuccess;
    if (a.dim % 4 || a.dim > 64 * 4 * MERGE_MAXV || a.k > BAND_MAX || !a.flag_count || !a.flag_list || !a.flag_tau || !a.flag2_count ||
        !a.flag2_list)
        return hipErrorInvalidValue;
    static unsigned long long attr = 0;     // bit d: set on device d
    set_max_dynamic_lds((const void*)band_select_kernel, BAND_MAX * 8, attr);
    // a fixed grid walks the flagged slots (two workgroups fit a CU: 64 KiB of keys each); all leave at once when nothing is flagged
    hipLaunchKernelGGL(band_select_kernel, dim3(max_slots < 512 ? max_slots : 512), dim3(256), BAND_MAX * 8, s, a, S, ldS, sub,
                       max_slots);
    return hipGetLastError();
}

}  // namespace vr
