// The exact fp32 pass behind vr_index_search's certification (search_common.h: certify_tail).
//
// The fused sweeps pick candidates by bf16-MFMA score and re-score them in fp32.  A query whose
// candidate lists cannot PROVE that no row outside them reaches the fp32 top-k (more near-ties around
// the k-th score than the lists hold, or a threshold that cut too high) is put on a flag list by its
// merge workgroup.  This file redoes those queries from the fp32 index alone:
//
//   exact_scores_kernel   S[slot][row] = fp32 dot(query flag_list[slot], row) for every row of the
//                         index, in the summation order of the re-scoring (dot_lane): a workgroup walks
//                         blocks of 64 rows; it keeps up to EX_QB flagged queries in LDS at a time and every wave
//                         takes 16 of the rows, a row's float4 chunks in registers.
//   bigk_select_kernel    (search_bigk.hip, exact mode) radix-selects the top k of each score row.
//
// Both launches are issued on EVERY search (the host cannot know the flag count without a
// synchronisation) and leave at once when nothing is flagged — the normal case; cost ~2 launch slots.
// Roofline when it does run: HBM (the fp32 index is read once per EX_QB flagged queries).
//
// Also here: the largest row norm and the largest bf16 rounding residual of the index rows (vr_index_add): the |d| and
// |d - bf16(d)| of the error bound.
#include "kernels.h"
#include "search_common.h"

namespace vr {

constexpr int EX_QB = 8;            // flagged queries resident in LDS per round (8 x dim x 4 B: 72 KiB at dim 2304)
constexpr int EX_ROWS = 64;         // rows per workgroup

__global__ __launch_bounds__(256) void exact_scores_kernel(const float* __restrict__ index_f32, int64_t n_docs, int dim,
                                                           const float* __restrict__ q_f32,
                                                           const int* __restrict__ flag_list,
                                                           const int* __restrict__ flag_count, int sub, int max_slots,
                                                           float* __restrict__ S, size_t ldS) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nf = min(flag_count[0] - sub, max_slots);     // this launch: entries [sub, sub + nf) of the list = slots 0..
    if (nf <= 0) return;
    flag_list += sub;
    f32x4* qs = reinterpret_cast<f32x4*>(smem);                 // [EX_QB][nv]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nv = dim >> 2;
    for (int f0 = 0; f0 < nf; f0 += EX_QB) {
        const int nb = min(EX_QB, nf - f0);
        __syncthreads();
        for (int e = tid; e < nb * nv; e += 256) {
            const int b = e / nv, c = e % nv;
            qs[b * nv + c] = reinterpret_cast<const f32x4*>(q_f32 + (size_t)flag_list[f0 + b] * dim)[c];
        }
        __syncthreads();
        // this wave's rows in order: 16 of every row block the workgroup walks; the next row's loads are in flight while
        // the current row is scored (two register sets, swapped by unrolling the loop twice)
        auto row_at = [&](int i) -> int64_t {
            return ((int64_t)blockIdx.x + (int64_t)(i >> 4) * gridDim.x) * EX_ROWS + wave * 16 + (i & 15);
        };
        auto load_row = [&](f32x4 (&dv)[MERGE_MAXV], int64_t row) {
            if (row >= n_docs) return;
            const f32x4* dr = reinterpret_cast<const f32x4*>(index_f32 + (size_t)row * dim);
#pragma unroll
            for (int i = 0; i < MERGE_MAXV; ++i) {
                const int c = lane + i * 64;
                dv[i] = (c < nv) ? dr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        auto score_row = [&](const f32x4 (&dv)[MERGE_MAXV], int64_t row) {
            for (int b = 0; b < nb; ++b) {
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < MERGE_MAXV; ++i) {
                    const int c = lane + i * 64;
                    if (c < nv) a = dot_chunk(qs[b * nv + c], dv[i], a);      // same chain as dot_lane(q, row)
                }
                a = wave_sum(a);
                if (lane == 0) S[(size_t)(f0 + b) * ldS + row] = a;
            }
        };
        f32x4 dA[MERGE_MAXV], dB[MERGE_MAXV];
        load_row(dA, row_at(0));
        for (int i = 0; row_at(i) < n_docs; i += 2) {             // (rows grow with i: wave-uniform exit)
            load_row(dB, row_at(i + 1));
            score_row(dA, row_at(i));
            if (row_at(i + 1) >= n_docs) break;
            load_row(dA, row_at(i + 2));
            score_row(dB, row_at(i + 1));
        }
    }
}

hipError_t launch_exact_scores(const float* index_f32, int64_t n_docs, int dim, const float* q_f32, const int* flag_list,
                               const int* flag_count, int sub, int max_slots, float* S, size_t ldS, hipStream_t s) {
    if (n_docs <= 0 || max_slots <= 0) return hipSuccess;
    if (dim % 4 || dim > 64 * 4 * MERGE_MAXV) return hipErrorInvalidValue;
    const int lds = EX_QB * dim * 4;
    static unsigned long long attr = 0;     // bit d: set on device d
    set_max_dynamic_lds((const void*)exact_scores_kernel, EX_QB * 64 * 4 * MERGE_MAXV * 4, attr);
    int64_t blocks = (n_docs + EX_ROWS - 1) / EX_ROWS;
    if (blocks > 512) blocks = 512;           // two workgroups per CU walk the row blocks (and leave at once when nothing is flagged)
    hipLaunchKernelGGL(exact_scores_kernel, dim3((unsigned)blocks), dim3(256), lds, s, index_f32, n_docs, dim, q_f32,
                       flag_list, flag_count, sub, max_slots, S, ldS);
    return hipGetLastError();
}

// ---- largest row norm ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_norm_max_kernel(const float* __restrict__ rows, int64_t n, int dim,
                                                           float* __restrict__ dmax) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = dim >> 2;
    float m = 0.f, mr = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n; r += (int64_t)gridDim.x * 4) {
        const f32x4* x = reinterpret_cast<const f32x4*>(rows + (size_t)r * dim);
        float ss = 0.f, rr = 0.f;
        for (int c = lane; c < nv; c += 64) {
            const f32x4 v = x[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = bf16_resid(v[e]);          // what launch_f32_to_bf16 drops from this element
                ss = __builtin_fmaf(v[e], v[e], ss);
                rr = __builtin_fmaf(d, d, rr);
            }
        }
        m = fmaxf(m, wave_sum(ss));
        mr = fmaxf(mr, wave_sum(rr));
    }
    // norms are >= 0: their bit patterns order like unsigned integers.  Round the roots up a little: bounds
    // (the fp32 sums of squares carry <= ~50 roundings of 2^-24 each).
    if (lane == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(dmax), __float_as_uint(sqrtf(m) * 1.00001f));
    if (lane == 0 && mr > 0.f) atomicMax(reinterpret_cast<unsigned*>(dmax) + 1, __float_as_uint(sqrtf(mr) * 1.00001f));
}

hipError_t launch_row_norm_max(const float* rows, int64_t n, int dim, float* dmax, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (dim % 4) return hipErrorInvalidValue;
    const int64_t nb4 = (n + 3) / 4;
    const int blocks = (int)(nb4 < 2048 ? nb4 : 2048);
    hipLaunchKernelGGL(row_norm_max_kernel, dim3(blocks), dim3(256), 0, s, rows, n, dim, dmax);
    return hipGetLastError();
}

// Worst case of |bf16-MFMA score - fp32 score| / (|q| |d|) (search_common.h): both operands rounded to nearest bf16 (8
// significand bits: unit roundoff 2^-8 each, 2^-7 + 2^-16 for the product), the exact products accumulated in fp32 over dim
// terms (<= 2^-23 per add, allowing a truncating adder), and the fp32 re-scoring's own rounding on the other side.  The
// default certification does not use this figure: it measures the rounding residuals of the data (query_eps).
float search_acc_rel(int dim) { return (float)(2 * dim + 128) * 0x1p-24f; }
float search_default_eps_rel(int dim) {
    return 0x1p-7f + 0x1p-16f + search_acc_rel(dim) * (1.f + 0x1p-8f) * (1.f + 0x1p-8f);
}

}  // namespace vr
