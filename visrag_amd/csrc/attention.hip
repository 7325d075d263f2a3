// Flash attention for the three attention shapes of the VisRAG-Ret encode path:
//   * SigLIP ViT self-attention, 16 heads x 72, non-causal, N = h*w patches
//       (F.scaled_dot_product_attention, vision_transformer.py:92-96)
//   * resampler cross-attention, 18 heads x 128, 64 batch-invariant queries x N keys
//       (nn.MultiheadAttention, resampler.py:128,158-163)
//   * MiniCPM decoder self-attention, 36 heads x 64, causal, ragged (packed) sequences
//       (MiniCPMSdpaAttention, modeling_minicpm.py:895-903; right padding never reaches a
//        valid row because causal already hides keys > query)
//
// One workgroup = 4 waves = QF*64 query rows of one (batch item, head); KV tiles of 64 keys
// staged through LDS; online softmax in fp32.  Both matmuls are TRANSPOSED so that the
// softmax statistics and P never leave registers:
//   S^T[key][q] = K * Q^T     (mfma A = K fragment from LDS, B = Q fragment kept in VGPRs)
//   O^T[d][q]  += V^T * P^T   (mfma A = V^T fragment from LDS, B = P^T = the S^T accumulator
//                              itself, re-packed to bf16 — the C layout of the first MFMA is
//                              a valid B layout of the second for a permuted key order, and
//                              V^T is stored in LDS in that same permuted order)
// A lane therefore owns one query column (q = lane & 15): max / sum are 16 in-lane values
// plus a 2-step cross-lane reduction, and the final O row is 4 consecutive d per lane
// (8-byte stores).  head_dim 72 is zero-padded to 96 for QK^T and to 80 for PV.
// Roofline: MFMA (4*N^2*D flop per head).
#include "common.h"
#include "kernels.h"

namespace vr {

template <int HD> struct AttnCfg;
template <> struct AttnCfg<64>  { static constexpr int KSTEPS = 2, DFRAGS = 4, KPITCH = 128; };
template <> struct AttnCfg<72>  { static constexpr int KSTEPS = 3, DFRAGS = 5, KPITCH = 256; };
template <> struct AttnCfg<128> { static constexpr int KSTEPS = 4, DFRAGS = 8, KPITCH = 256; };

constexpr int ATT_KV = 64;          // keys per tile

// position of key `key` (0..63) inside a V^T row: MFMA k-slot order of the PV product
__device__ __forceinline__ int vt_slot(int key) {
    return (key & 32) | (((key >> 2) & 3) << 3) | (((key >> 4) & 1) << 2) | (key & 3);
}

template <int HD, int QF>
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs p) {
    using C = AttnCfg<HD>;
    constexpr int KSTEPS = C::KSTEPS, DFRAGS = C::DFRAGS, KPITCH = C::KPITCH;
    constexpr int KCH = KPITCH / 16;              // 16-byte chunks per K row in LDS
    constexpr int CPR = HD / 8;                   // 16-byte chunks per global row
    constexpr int QT = 64 * QF;                   // query rows per workgroup
    constexpr int NCH = (ATT_KV * CPR + 255) / 256;   // staging chunks per thread

    __shared__ __attribute__((aligned(16))) char smem[ATT_KV * KPITCH + DFRAGS * 16 * 128];
    char* Ks = smem;
    char* Vt = smem + ATT_KV * KPITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;

    const int q_tiles = (p.max_q + QT - 1) / QT;
    const int t = xcd_remap(blockIdx.x, p.B * p.heads * q_tiles);
    const int qt = t % q_tiles, bh = t / q_tiles;
    const int h = bh % p.heads, b = bh / p.heads;

    const int kv0 = p.cu_kv[b], kv_len = p.cu_kv[b + 1] - kv0;
    const int q_row0 = p.cu_q[b];                      // rows of `out` (and of q unless shared)
    const int q_len = p.cu_q[b + 1] - q_row0;
    const int qs = qt * QT;                            // first query of this tile (seq-relative)
    if (qs >= q_len) return;

    const bf16_t* qbase = (const bf16_t*)p.q + (size_t)(p.q_shared ? 0 : q_row0) * p.ldq + h * HD;
    const bf16_t* kbase = (const bf16_t*)p.k + (size_t)kv0 * p.ldk + h * HD;
    const bf16_t* vbase = (const bf16_t*)p.v + (size_t)kv0 * p.ldv + h * HD;

    // ---- zero the LDS padding that staging never overwrites (K pad columns, V^T pad rows)
    for (int i = tid; i < (int)sizeof(smem) / 16; i += 256)
        reinterpret_cast<u32x4*>(smem)[i] = u32x4{0, 0, 0, 0};

    // ---- Q fragments (B operand of S^T): lane holds Q[q = fr][d = ks*32 + fq*8 .. +7]
    bf16x8 qf[QF][KSTEPS];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const int q = qs + (wave * QF + f) * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int d = ks * 32 + fq * 8;
            u32x4 raw = {0, 0, 0, 0};
            if (q < q_len && d < HD) raw = *reinterpret_cast<const u32x4*>(qbase + (size_t)q * p.ldq + d);
            qf[f][ks] = __builtin_bit_cast(bf16x8, raw);
        }
    }

    f32x4 o[QF][DFRAGS];
    float m_run[QF], l_run[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        m_run[f] = -INFINITY; l_run[f] = 0.f;
#pragma unroll
        for (int d = 0; d < DFRAGS; ++d) o[f][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    int kv_end = kv_len;
    if (p.causal) kv_end = min(kv_len, qs + QT);
    const int n_tiles = (kv_end + ATT_KV - 1) / ATT_KV;
    const float sc = p.scale * 1.44269504088896340736f;   // exp2 domain

    u32x4 rk[NCH], rv[NCH];
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * 256;
            const int key = c / CPR, ch = c % CPR;
            const int kg = tile * ATT_KV + key;
            rk[i] = u32x4{0, 0, 0, 0}; rv[i] = u32x4{0, 0, 0, 0};
            if (c < ATT_KV * CPR && kg < kv_len) {
                rk[i] = *reinterpret_cast<const u32x4*>(kbase + (size_t)kg * p.ldk + ch * 8);
                rv[i] = *reinterpret_cast<const u32x4*>(vbase + (size_t)kg * p.ldv + ch * 8);
            }
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * 256;
            if (c >= ATT_KV * CPR) continue;
            const int key = c / CPR, ch = c % CPR;
            // K row-major, chunk XOR-swizzled: conflict-free ds_read_b128 of the A fragments
            *reinterpret_cast<u32x4*>(Ks + key * KPITCH + ((ch ^ (key & (KCH - 1))) << 4)) = rk[i];
            // V transposed: Vt[d][slot(key)], 8-chunk rows swizzled by (d & 7)
            const int pos = vt_slot(key);
            const bf16x8 vv = __builtin_bit_cast(bf16x8, rv[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = ch * 8 + e;
                *reinterpret_cast<bf16_t*>(Vt + d * 128 + ((((pos >> 3) ^ (d & 7)) << 4) | ((pos & 7) << 1))) = vv[e];
            }
        }
    };

    load_tile(0);
    for (int tile = 0; tile < n_tiles; ++tile) {
        __syncthreads();            // previous tile fully consumed (and the zero-fill done)
        write_tile();
        __syncthreads();
        if (tile + 1 < n_tiles) load_tile(tile + 1);   // in flight during the MFMAs below

        const int key0 = tile * ATT_KV;
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            // ---- S^T = K Q^T : 4 key fragments x KSTEPS
            f32x4 s[4];
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
                s[kf] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int row = kf * 16 + fr;
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    const bf16x8 ka = *reinterpret_cast<const bf16x8*>(
                        Ks + row * KPITCH + (((ks * 4 + fq) ^ (row & (KCH - 1))) << 4));
                    s[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[f][ks], s[kf], 0, 0, 0);
                }
            }
            // ---- mask + online softmax; this lane: query q, keys key0 + kf*16 + fq*4 + r
            const int q = qs + (wave * QF + f) * 16 + fr;
            const int lim = p.causal ? min(kv_len - 1, q) : kv_len - 1;   // last visible key
            float mx = -INFINITY;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = key0 + kf * 16 + fq * 4 + r;
                    const float v = (key <= lim) ? s[kf][r] * sc : -INFINITY;
                    s[kf][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[f], mx);
            // rows with no visible key so far keep m = -inf; use 0 there so exp2(-inf - 0) = 0
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = exp2f(m_run[f] - m_use);
            float rs = 0.f;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = exp2f(s[kf][r] - m_use);
                    s[kf][r] = e;
                    rs += e;
                }
            rs += __shfl_xor(rs, 16, 64);
            rs += __shfl_xor(rs, 32, 64);
            l_run[f] = l_run[f] * alpha + rs;
            m_run[f] = m_new;
#pragma unroll
            for (int d = 0; d < DFRAGS; ++d) o[f][d] *= alpha;
            // ---- O^T += V^T P^T : P^T fragment for k-step ks = S^T fragments (2ks, 2ks+1)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 pb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pb[r] = f2bf(s[2 * ks][r]);
                    pb[4 + r] = f2bf(s[2 * ks + 1][r]);
                }
#pragma unroll
                for (int d = 0; d < DFRAGS; ++d) {
                    const int row = d * 16 + fr;
                    const bf16x8 va = *reinterpret_cast<const bf16x8*>(
                        Vt + row * 128 + (((ks * 4 + fq) ^ (row & 7)) << 4));
                    o[f][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, o[f][d], 0, 0, 0);
                }
            }
        }
    }

    // ---- normalise and store: lane owns out[q][h*HD + d*16 + fq*4 .. +3]
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const int q = qs + (wave * QF + f) * 16 + fr;
        if (q >= q_len) continue;
        const float inv = 1.0f / l_run[f];
        bf16_t* orow = (bf16_t*)p.out + (size_t)(q_row0 + q) * p.ldo + h * HD;
#pragma unroll
        for (int d = 0; d < DFRAGS; ++d) {
            const int dd = d * 16 + fq * 4;
            if (dd < HD) {
                bf16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = f2bf(o[f][d][r] * inv);
                *reinterpret_cast<bf16x4*>(orow + dd) = ov;
            }
        }
    }
}

template <int HD, int QF>
static hipError_t launch_t(const AttnArgs& a, hipStream_t s) {
    const int q_tiles = (a.max_q + 64 * QF - 1) / (64 * QF);
    hipLaunchKernelGGL((attention_kernel<HD, QF>), dim3(a.B * a.heads * q_tiles), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
    if (a.B <= 0 || a.max_q <= 0) return hipSuccess;
    if ((a.ldq | a.ldk | a.ldv) % 8 || a.ldo % 4) return hipErrorInvalidValue;
    const bool big = a.max_q > 64;
    switch (a.head_dim) {
        case 64:  return big ? launch_t<64, 2>(a, s) : launch_t<64, 1>(a, s);
        case 72:  return big ? launch_t<72, 2>(a, s) : launch_t<72, 1>(a, s);
        case 128: return launch_t<128, 1>(a, s);
        default:  return hipErrorInvalidValue;
    }
}

}  // namespace vr
