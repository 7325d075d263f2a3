// bf16 GEMM for M <= 16 rows, second form: the weights go from HBM straight into REGISTERS.
//
//   out[split][m][n] = A[m, Ks] * W[n, Ks]^T      (same contract as gemm_skinny.hip; launch_gemm_skinny picks the form)
//
// gemm_skinny.hip streams W through four LDS stages: 96 KiB of requests in flight per CU is all that 160 KiB of LDS allow,
// and at ~2.6 us of loaded HBM latency that is 5.4 TB/s on the 148 workgroups of the 7B gate / up projection (14 MB in
// flight chip-wide; the lm_head's 594 workgroups reach 6.2).  The decode step is a stream of such launches, so the bytes in
// flight are the lever, and the register file is the bigger buffer: 512 KiB per CU.  Here a workgroup is EIGHT waves x 32
// output columns (the same 256-column tile and K split as the other form); a wave requests its 32 rows x 64 k slabs as
// four loads of 8 rows x 128 B (whole cache lines, the LDS-DMA's pattern; lanes fetching the MFMA operand directly — 16
// rows x 64 B per instruction — measured 3.6-4.1 TB/s) TWELVE slabs ahead: 48 loads = 48 KiB per wave, 384 KiB per CU in
// flight, four times the LDS ring; a landed slab crosses a 4 KiB wave-private LDS pad into fragment layout.  The K range's activations (M <= 16 rows, at most 64 slabs) are staged ONCE
// in LDS (row pitch + 16 B: conflict-free fragment reads) — requested before the weight burst, because a wave's loads
// return in order and the first MFMA needs them.  Roofline: HBM (N * K * 2 bytes of weights per launch).
#include "gemm_core.h"
#include "kernels.h"

namespace vr {

namespace {

constexpr int ST_U = 12;                   // K slabs (64 k) of weights in flight per wave
constexpr int ST_MAX_SLABS = 60;           // K range of a workgroup: 16 rows x (60 x 128 + 16) B of LDS + 8 x 4 KiB of transposition pads
constexpr int ST_PAD = 32 * 128;           // a wave's 32 rows x 64 k of the slab it is about to multiply
constexpr unsigned ST_OOB = 0x80000000u;
#ifndef VR_STREAM_W_AUX
#define VR_STREAM_W_AUX 2
#endif
constexpr int ST_W_AUX = VR_STREAM_W_AUX;   // nt: every byte of W is read once, by one CU

}  // namespace

int gemm_stream_max_slabs() { return ST_MAX_SLABS; }

// SWIGLU / COMBINE: as in gemm_skinny.hip (COMBINE: any K range that fits the LDS rows)
template <bool SWIGLU, bool COMBINE>
__global__ __launch_bounds__(512) void gemm_stream_kernel(GemmArgs p, SkinnyCombine cb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = (p.N + 255) / 256;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int split = blockIdx.x / tiles_n, tn = blockIdx.x - split * tiles_n;
    const int n0 = tn * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int nk_all = p.K / GEMM_BK, per = (nk_all + ks - 1) / ks;
    const int k0 = split * per, nk = max(0, min(per, nk_all - k0));
    const size_t kof = (size_t)k0 * GEMM_BK;
    const int pitch = per * 128 + 16;                       // bytes of an A row in LDS

    // ---- activations: requested first
    const int a_chunks = p.M * nk * 8;                      // 16-byte chunks of the K range's A rows
    const bool a_early = a_chunks <= 1024;                  // <= 2 per thread: held in registers under the weight burst
    u32x4 ar[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
    float cl[GEN_ATT_SPLITS], cpv[GEN_ATT_SPLITS];
    int cS = 0;
    if constexpr (COMBINE) {
        // (M = 1, nk * 64 <= 512 columns: one column per thread; ranges of the decode attention merged as in gemm_skinny.hip)
        cS = cb.S_dev ? *cb.S_dev : cb.S;
        const int col = (int)kof + min(tid, max(nk * 64 - 1, 0)), h = col >> 7, d = col & 127;
        const int kvh = cb.heads / cb.group, hkv = h / cb.group, g = h % cb.group;
#pragma unroll
        for (int t = 0; t < GEN_ATT_SPLITS; ++t) {
            const int r = (min(t, max(cS - 1, 0)) * cb.group + g) * kvh + hkv;
            cl[t] = cb.lse[r];
            cpv[t] = bf2f(((const bf16_t*)cb.part)[(size_t)r * 128 + d]);
        }
#pragma unroll
        for (int t = 0; t < GEN_ATT_SPLITS; ++t)
            if (t >= cS) cl[t] = -INFINITY;
    } else {
        auto a_src = [&](int c) {                           // chunk c -> row c / (nk * 8), 16-byte chunk c % (nk * 8)
            const int m = c / (nk * 8), ch = c - m * (nk * 8);
            return (const char*)p.A + ((size_t)m * p.lda + kof) * 2 + (size_t)ch * 16;
        };
        if (a_early) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (tid + i * 512 < a_chunks) ar[i] = *reinterpret_cast<const u32x4*>(a_src(tid + i * 512));
        } else {
            for (int c = tid; c < a_chunks; c += 512) {
                const int m = c / (nk * 8), ch = c - m * (nk * 8);
                *reinterpret_cast<u32x4*>(smem + m * pitch + ch * 16) = *reinterpret_cast<const u32x4*>(a_src(c));
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);                      // (hipcc would hoist the weight requests above them)

    // ---- the weight burst: ST_U slabs x 4 loads of 8 rows x 128 B (whole cache lines: 8 lanes per row)
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.W + ((size_t)(n0 + wave * 32) * p.ldw + kof) * 2), 0,
                                                         0x7FFFFFFF, 0x00020000);
    const int r8 = lane >> 3, c8 = lane & 7;
    const unsigned lof = (unsigned)r8 * (unsigned)p.ldw * 2u + (unsigned)c8 * 16u;
    const unsigned rg = 8u * (unsigned)p.ldw * 2u;
    u32x4 w[ST_U][4];
    auto issue = [&](int u, int s) {
        const unsigned so = s < nk ? (unsigned)s * 128u : ST_OOB;           // past the end: out of the descriptor's range, zeros, no traffic
#pragma unroll
        for (int i = 0; i < 4; ++i) w[u][i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, lof + i * rg, so, ST_W_AUX);
    };
#pragma unroll
    for (int u = 0; u < ST_U; ++u) issue(u, u);
    __builtin_amdgcn_sched_barrier(0);

    // ---- A rows into LDS (the tail of a short last split stays unread: the MFMAs below stop at nk)
    if constexpr (COMBINE) {
        if (tid < nk * 64) {
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < GEN_ATT_SPLITS; ++t) mx = fmaxf(mx, cl[t]);
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int t = 0; t < GEN_ATT_SPLITS; ++t) {
                const float e = exp2f(cl[t] - mx);              // (ranges past the last: 2^-inf = 0)
                num += e * cpv[t];
                den += e;
            }
            *reinterpret_cast<bf16_t*>(smem + tid * 2) = f2bf(num / den);
        }
    } else if (a_early) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * 512;
            if (c < a_chunks) {
                const int m = c / (nk * 8), ch = c - m * (nk * 8);
                *reinterpret_cast<u32x4*>(smem + m * pitch + ch * 16) = ar[i];
            }
        }
    }
    __syncthreads();

    f32x4 acc[2][2];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[f][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* arow = smem + fr * pitch + fq * 16;          // rows >= M: whatever the LDS holds — they feed output rows nobody stores
    // the wave's transposition pad (its own 4 KiB: no barrier, a wave's LDS operations execute in order): rows of 128 B with
    // gemm_core.h's chunk swizzle — written as loaded (lane = row r8 of the group, chunk c8), read as MFMA fragments
    char* pad = smem + 16 * pitch + wave * ST_PAD;
    char* pw = pad + r8 * 128 + ((c8 ^ r8) << 4);
    const char* pr0 = pad + fr * 128 + ((fq ^ (fr & 7)) << 4), *pr1 = pad + fr * 128 + (((4 + fq) ^ (fr & 7)) << 4);
    for (int s0 = 0; s0 < nk; s0 += ST_U) {
#pragma unroll
        for (int u = 0; u < ST_U; ++u) {
            const int s = s0 + u;
            if (s < nk) {
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(pw + i * 1024) = w[u][i];
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(arow + s * 128), a1 = *reinterpret_cast<const bf16x8*>(arow + s * 128 + 64);
                const bf16x8 w00 = *reinterpret_cast<const bf16x8*>(pr0), w01 = *reinterpret_cast<const bf16x8*>(pr1);
                const bf16x8 w10 = *reinterpret_cast<const bf16x8*>(pr0 + 2048), w11 = *reinterpret_cast<const bf16x8*>(pr1 + 2048);
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w00, a0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w01, a1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w10, a0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w11, a1, acc[1][1], 0, 0, 0);
            }
            issue(u, s + ST_U);
        }
    }
    // out[m = fr][n = n0 + wave*32 + f*16 + fq*4 + r]; bias rides with split 0
    const f32x4 o0 = acc[0][0] + acc[0][1], o1 = acc[1][0] + acc[1][1];
    if (fr >= p.M) return;
    if constexpr (SWIGLU) {
        const int n = n0 + wave * 32;                        // a [16 gate | 16 up] block pair = 16 columns of act
        if (n < p.N) {
            f32x4 g = o0, uu = o1;
            if (p.bias) {
                g += *reinterpret_cast<const f32x4*>(p.bias + n + fq * 4);
                uu += *reinterpret_cast<const f32x4*>(p.bias + n + 16 + fq * 4);
            }
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(g[r] / (1.0f + __expf(-g[r])) * uu[r]);   // (swiglu_sum_kernel's expression)
            *reinterpret_cast<bf16x4*>((bf16_t*)p.out + (size_t)fr * p.ldo + n / 2 + fq * 4) = o;
        }
        return;
    }
    float* out = (float*)p.out + (size_t)split * p.split_stride + (size_t)fr * p.ldo;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int n = n0 + wave * 32 + f * 16 + fq * 4;
        if (n < p.N) {
            f32x4 v = f ? o1 : o0;
            if (p.bias && split == 0) v += *reinterpret_cast<const f32x4*>(p.bias + n);
            *reinterpret_cast<f32x4*>(out + n) = v;
        }
    }
}

// the caller (launch_gemm_skinny) has checked the shared contract; here: the K range fits the LDS rows
hipError_t launch_gemm_stream(const GemmArgs& a, hipStream_t s, bool swiglu, const SkinnyCombine* combine) {
    const int ks = a.ksplit > 1 ? a.ksplit : 1;
    const int nk_all = a.K / GEMM_BK, per = (nk_all + ks - 1) / ks;
    if (per > ST_MAX_SLABS) return hipErrorInvalidValue;
    if (combine && (a.M != 1 || per * 64 > 512)) return hipErrorInvalidValue;
    const size_t tn = (a.N + 255) / 256;
    const int smem = 16 * (per * 128 + 16) + 8 * ST_PAD;
    static unsigned long long attr = 0, attr_sw = 0, attr_cb = 0;     // bit d: set on device d
    if (combine) {
        set_max_dynamic_lds((const void*)gemm_stream_kernel<false, true>, 16 * (ST_MAX_SLABS * 128 + 16) + 8 * ST_PAD, attr_cb);
        hipLaunchKernelGGL((gemm_stream_kernel<false, true>), dim3((unsigned)(tn * ks)), dim3(512), smem, s, a, *combine);
    } else if (swiglu) {
        set_max_dynamic_lds((const void*)gemm_stream_kernel<true, false>, 16 * (ST_MAX_SLABS * 128 + 16) + 8 * ST_PAD, attr_sw);
        hipLaunchKernelGGL((gemm_stream_kernel<true, false>), dim3((unsigned)tn), dim3(512), smem, s, a, SkinnyCombine{});
    } else {
        set_max_dynamic_lds((const void*)gemm_stream_kernel<false, false>, 16 * (ST_MAX_SLABS * 128 + 16) + 8 * ST_PAD, attr);
        hipLaunchKernelGGL((gemm_stream_kernel<false, false>), dim3((unsigned)(tn * ks)), dim3(512), smem, s, a, SkinnyCombine{});
    }
    return hipGetLastError();
}

}  // namespace vr
