// bf16 GEMM for M <= 16 rows (MB = 2: M <= 32): the generator's decode step and the retriever's single-query path,
// where a GEMM is a pass over the weights.
//
//   out[split][m][n] = A[m, Ks] * W[n, Ks]^T      (fp32 partial planes, summed by their consumer)
//
// The 256 x 256 tiles of the other kernels spend a K-step (64 k) on 128 MFMAs per wave of which one row in sixteen
// is real, and stream their 32 KiB of weights per step at the pace of that arithmetic: 26 GB/s per CU, 1.6 TB/s
// for the 7B decoder.  Here a workgroup owns 256 output columns and a K range, four waves x 64 columns, EIGHT
// MFMAs per wave and K-step, and four LDS stages (32 KiB of W + 2 KiB of A each) keep three K-steps of LDS-DMA in
// flight: ~100 KiB per CU against ~2 us of HBM latency, so the chip's 256 CUs can ask for more than the HBM
// delivers.  Roofline: HBM (N * K * 2 bytes of weights per launch).
//
// LDS image and swizzle of gemm_core.h (16-byte chunk c of row r at chunk c ^ (r & 7)); every wave issues the same
// nine loads per K-step (eight 8-row groups of its 64 W rows + one 8-row group of A — waves 2 and 3 repeat the
// groups of waves 0 and 1) so one `vmcnt(18)` fits all; K-steps past the end point out of the descriptor's range.
#include "gemm_core.h"
#include "gen_math.h"
#include "kernels.h"

namespace vr {

namespace {

constexpr int SK_STAGES = 4;
constexpr int SK_W_BYTES = 256 * 128;
constexpr int sk_stage(int mb) { return SK_W_BYTES + mb * 16 * 128; }      // 32 KiB of W + 2 KiB of A per 16 rows
constexpr int sk_smem(int mb) { return SK_STAGES * sk_stage(mb); }
constexpr unsigned SK_OOB = 0x80000000u;
// cache policy of the weight loads: nt (aux bit 1) — every byte of W is read once, by one CU
#ifndef VR_SKINNY_W_AUX
#define VR_SKINNY_W_AUX 2
#endif
constexpr int SK_W_AUX = VR_SKINNY_W_AUX;

}  // namespace

// SWIGLU (ksplit 1 only): W rows interleaved in blocks of 16 ([16 gate | 16 up | ...], EPI_SWIGLU's layout); the tile's
// epilogue writes act = silu(gate) * up as bf16 [M][ldo] — no fp32 plane, no swiglu_sum launch.
// COMBINE (the decode step's o projection; M = 1, at most SK_STAGES K-steps per workgroup): A is not read from memory —
// row 0 of every K-step's A stage is built in the prologue from the partial attention rows of the step's KV ranges,
//     A[col] = sum_s 2^(lse_s - max) part_s[col] / sum_s 2^(lse_s - max)   per query head (attn_combine_kernel's sum; layout: kernels.h)
// one lane per column, one wave per K-step: the launch that used to merge the ranges is gone.
// MB = 2 (plain planes only): 32 rows per pass — two 16-row MFMA row blocks share every W fragment; the A stage has four
// 8-row groups, one per wave (same nine loads per wave and K-step).
template <bool SWIGLU, bool COMBINE, int MB = 1>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p, SkinnyCombine cb) {
    static_assert(MB == 1 || (MB == 2 && !SWIGLU && !COMBINE), "32-row passes write plain fp32 planes");
    constexpr int SK_STAGE = sk_stage(MB);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = (p.N + 255) / 256;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int split = blockIdx.x / tiles_n, tn = blockIdx.x - split * tiles_n;
    const int n0 = tn * 256;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    // K-steps of this split: ceil(steps / ks) each, the last split takes what is left (ks need not divide the steps)
    const int nk_all = p.K / GEMM_BK, per = (nk_all + ks - 1) / ks;
    const int k0 = split * per, nk = max(0, min(per, nk_all - k0));
    const size_t kof = (size_t)k0 * GEMM_BK;

    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.W + ((size_t)n0 * p.ldw + kof) * 2), 0,
                                                         0x7FFFFFFF, 0x00020000);
    const auto arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A + kof * 2), 0, 0x7FFFFFFF, 0x00020000);
    const unsigned lchunk = (unsigned)(((lane & 7) ^ (lane >> 3)) << 4);
    const unsigned lofW = (unsigned)(lane >> 3) * (unsigned)p.ldw * 2u + lchunk;
    const unsigned lofA = (unsigned)(lane >> 3) * (unsigned)p.lda * 2u + lchunk;
    const unsigned rgW = (unsigned)p.ldw * 16u;
    const int agrp = MB == 2 ? wave : (wave & 1);                     // this wave's 8-row group of the A stage
    const unsigned sW0 = (unsigned)wave * 8u * rgW, sA0 = (unsigned)agrp * (unsigned)p.lda * 16u;
    auto issue = [&](int kt) {                      // the 9 loads of K-step kt into stage kt % 4
        char* st = smem + (kt & (SK_STAGES - 1)) * SK_STAGE;
        const unsigned kb = kt < nk ? (unsigned)kt * (GEMM_BK * 2) : SK_OOB;
#pragma unroll
        for (int d = 0; d < 8; ++d)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, VR_LDS(st + wave * 8192 + d * 1024), 16, lofW + kb, sW0 + d * rgW, 0, SK_W_AUX);
        if constexpr (!COMBINE)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, VR_LDS(st + SK_W_BYTES + agrp * 1024), 16, lofA + kb, sA0, 0, 0);
    };
    f32x4 acc[MB][4];
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[b][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // COMBINE: the partial rows and their log-sum-exps are requested BEFORE the weight stream (loads return in order: asked
    // for after it, they would wait behind three K-steps of weights), merged while the weights are in flight
    float cl[GEN_ATT_SPLITS], cpv[GEN_ATT_SPLITS];
    int cS = 0;
    if constexpr (COMBINE) {
        cS = cb.S_dev ? *cb.S_dev : cb.S;
        const int col = (int)kof + min(wave, max(nk - 1, 0)) * GEMM_BK + lane, h = col >> 7, d = col & 127;
        const int kvh = cb.heads / cb.group, hkv = h / cb.group, g = h % cb.group;
        // (unconditional loads at a clamped range index: sixteen independent requests in flight, not sixteen round trips)
#pragma unroll
        for (int t = 0; t < GEN_ATT_SPLITS; ++t) {
            const int r = (min(t, max(cS - 1, 0)) * cb.group + g) * kvh + hkv;
            cl[t] = cb.lse[r];
            cpv[t] = bf2f(((const bf16_t*)cb.part)[(size_t)r * 128 + d]);
        }
#pragma unroll
        for (int t = 0; t < GEN_ATT_SPLITS; ++t)
            if (t >= cS) cl[t] = -INFINITY;
    }
    if constexpr (COMBINE) __builtin_amdgcn_sched_barrier(0);        // (hipcc would hoist the weight requests above them)
    issue(0); issue(1); issue(2);
    if constexpr (COMBINE) {
        // wave w builds row 0 of K-step w's A stage (nk <= SK_STAGES: every step has its own stage, nothing recycles it);
        // rows 1..15 of the stage belong to output rows that are never stored
        if (wave < nk) {
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < GEN_ATT_SPLITS; ++t) mx = fmaxf(mx, cl[t]);
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int t = 0; t < GEN_ATT_SPLITS; ++t) {
                const float w = exp2f(cl[t] - mx);              // (ranges past the last: 2^-inf = 0)
                merge_range(num, den, w, cpv[t]);
            }
            *reinterpret_cast<bf16_t*>(smem + wave * SK_STAGE + SK_W_BYTES + lane * 2) = f2bf(num / den);
        }
        // (the first barrier of the loop below orders these LDS writes before every wave's reads)
    }
    const int ch0 = (fq ^ (fr & 7)) << 4, ch1 = ((4 + fq) ^ (fr & 7)) << 4;
    for (int kt = 0; kt < nk; ++kt) {
        if constexpr (COMBINE) { asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
        else { VR_WAIT_VM_BARRIER(18); }            // K-step kt has landed everywhere; everyone is done with stage (kt - 1) % 4
        issue(kt + 3);
        const char* st = smem + (kt & (SK_STAGES - 1)) * SK_STAGE;
        const char* wr = st + (wave * 64 + fr) * 128;
        const char* ar = st + SK_W_BYTES + fr * 128;
        bf16x8 a0[MB], a1[MB];
#pragma unroll
        for (int b = 0; b < MB; ++b) {
            a0[b] = *reinterpret_cast<const bf16x8*>(ar + b * 2048 + ch0);
            a1[b] = *reinterpret_cast<const bf16x8*>(ar + b * 2048 + ch1);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wr + j * 2048 + ch0);
            const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wr + j * 2048 + ch1);
#pragma unroll
            for (int b = 0; b < MB; ++b) {
                acc[b][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, a0[b], acc[b][j], 0, 0, 0);
                acc[b][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, a1[b], acc[b][j], 0, 0, 0);
            }
        }
    }
    // acc[b][j][r] = out[m = b*16 + fr][n = n0 + wave*64 + j*16 + fq*4 + r]; bias rides with split 0
    if constexpr (SWIGLU) {
        if (fr < p.M) {
            bf16_t* act = (bf16_t*)p.out + (size_t)fr * p.ldo;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int n = n0 + wave * 64 + jj * 32;          // a [16 gate | 16 up] block pair = 16 columns of act
                if (n < p.N) {
                    f32x4 g = acc[0][2 * jj], u = acc[0][2 * jj + 1];
                    if (p.bias) {
                        g += *reinterpret_cast<const f32x4*>(p.bias + n + fq * 4);
                        u += *reinterpret_cast<const f32x4*>(p.bias + n + 16 + fq * 4);
                    }
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = f2bf(g[r] / (1.0f + __expf(-g[r])) * u[r]);   // (swiglu_sum_kernel's expression)
                    *reinterpret_cast<bf16x4*>(act + n / 2 + fq * 4) = o;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < MB; ++b) {
        const int m = b * 16 + fr;
        if (m < p.M) {
            float* out = (float*)p.out + (size_t)split * p.split_stride + (size_t)m * p.ldo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wave * 64 + j * 16 + fq * 4;
                if (n < p.N) {
                    f32x4 v = acc[b][j];
                    if (p.bias && split == 0) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                    *reinterpret_cast<f32x4*>(out + n) = v;
                }
            }
        }
    }
}

// fp32 planes out[split][M][ldo]; M <= 16, or <= 32 for plain planes (A and W rows readable up to 16 resp. 32 / the next
// multiple of 256), K % 64 == 0; split s
// covers K-steps [s * ceil(steps / ksplit), ...) — a split past the end writes a plane of zeros (+ bias for split 0)
hipError_t launch_gemm_skinny(const GemmArgs& a, hipStream_t s, bool swiglu, const SkinnyCombine* combine) {
    const int ks = a.ksplit > 1 ? a.ksplit : 1;
    if (a.M <= 0) return hipSuccess;
    if (combine) {
        // one row, every workgroup's K range inside the stage ring, columns = heads x 128
        const int nk_all = a.K / GEMM_BK, per = (nk_all + ks - 1) / ks;
        if (swiglu || a.M != 1 || per > SK_STAGES || a.K != combine->heads * 128 || a.K % GEMM_BK || combine->group <= 0 ||
            combine->heads % combine->group) return hipErrorInvalidValue;
    }
    if (swiglu && (ks != 1 || a.N % 32)) return hipErrorInvalidValue;
    const bool wide = a.M > 16;
    if (a.M > 32 || (wide && (swiglu || combine)) || a.N % 4 || a.K % GEMM_BK || a.rowmap || a.rowbias) return hipErrorInvalidValue;
    const size_t tn = (a.N + 255) / 256;
    if (tn * 256 * (size_t)a.ldw * 2 >= (1ull << 31) || 32 * (size_t)a.lda * 2 >= (1ull << 31)) return hipErrorInvalidValue;
    static unsigned long long attr = 0, attr_sw = 0;     // bit d: set on device d
    static unsigned long long attr_cb = 0, attr_w = 0;
    constexpr int SK_SMEM = sk_smem(1);
    if (combine) {
        set_max_dynamic_lds((const void*)gemm_skinny_kernel<false, true>, SK_SMEM, attr_cb);
        hipLaunchKernelGGL((gemm_skinny_kernel<false, true>), dim3((unsigned)(tn * ks)), dim3(256), SK_SMEM, s, a, *combine);
    } else if (swiglu) {
        set_max_dynamic_lds((const void*)gemm_skinny_kernel<true, false>, SK_SMEM, attr_sw);
        hipLaunchKernelGGL((gemm_skinny_kernel<true, false>), dim3((unsigned)tn), dim3(256), SK_SMEM, s, a, SkinnyCombine{});
    } else if (wide) {
        set_max_dynamic_lds((const void*)gemm_skinny_kernel<false, false, 2>, sk_smem(2), attr_w);
        hipLaunchKernelGGL((gemm_skinny_kernel<false, false, 2>), dim3((unsigned)(tn * ks)), dim3(256), sk_smem(2), s, a, SkinnyCombine{});
    } else {
        set_max_dynamic_lds((const void*)gemm_skinny_kernel<false, false>, SK_SMEM, attr);
        hipLaunchKernelGGL((gemm_skinny_kernel<false, false>), dim3((unsigned)(tn * ks)), dim3(256), SK_SMEM, s, a, SkinnyCombine{});
    }
    return hipGetLastError();
}

// act[m][i] = silu(gate) * up from the fp32 partial planes of a gate/up GEMM whose W rows are interleaved in blocks
// of 16 ([16 gate | 16 up | ...], EPI_SWIGLU's layout): gate of column i sits at (i / 16) * 32 + i % 16, up 16 further
__global__ void swiglu_sum_kernel(const float* __restrict__ parts, int n_parts, size_t plane_stride, int ldp, int M, int I,
                                  bf16_t* __restrict__ act, int lda) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (i >= I) return;
    const size_t g = (size_t)m * ldp + (size_t)(i >> 4) * 32 + (i & 15);
    float gate = 0.f, up = 0.f;
    for (int sp = 0; sp < n_parts; ++sp) { gate += parts[sp * plane_stride + g]; up += parts[sp * plane_stride + g + 16]; }
    act[(size_t)m * lda + i] = f2bf(gate / (1.0f + __expf(-gate)) * up);
}
hipError_t launch_swiglu_sum(const float* parts, int n_parts, size_t plane_stride, int ldp, int M, int I, void* act, int lda,
                             hipStream_t s) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(swiglu_sum_kernel, dim3((I + 255) / 256, M), dim3(256), 0, s, parts, n_parts, plane_stride, ldp, M, I,
                       (bf16_t*)act, lda);
    return hipGetLastError();
}

}  // namespace vr
