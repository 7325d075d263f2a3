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
// staged ROW-MAJOR through LDS (16-byte writes); online softmax in fp32.  Both matmuls are
// TRANSPOSED so that the softmax statistics and P never leave registers:
//   S^T[key][q] = K * Q^T     (mfma A = K fragment, ds_read_b128; B = Q fragment kept in VGPRs)
//   O^T[d][q]  += V^T * P^T   (mfma A = V^T fragment, fetched with ds_read_b64_tr_b16 — the
//                              gfx950 transposing LDS read — straight from the row-major V
//                              tile; B = P^T = the S^T accumulator itself re-packed to bf16:
//                              the C layout of the first MFMA is a valid B layout of the second
//                              for the key order (2ks*16 + g*4 + j | (2ks+1)*16 + g*4 + j), and
//                              the two tr-reads of a V fragment fetch exactly those keys)
// A lane therefore owns one query column (q = lane & 15): max / sum are 16 in-lane values plus
// a 2-step cross-lane reduction, and the final O row is 4 consecutive d per lane (8-byte stores).
// head_dim 72: QK^T = 2 x (16x16x32) + 1 x (16x16x16) MFMA (d padded to 80 with zeros in LDS),
// PV = 5 d-fragments (80).  Row pitch 160 B (288 B for head_dim 128) = 8 mod 64 dwords makes
// both the b128 K reads and the tr_b16 V reads bank-conflict free.
// Roofline: MFMA (4*N^2*D flop per head).
#include "common.h"
#include "kernels.h"

#ifndef VR_ATTN_TAIL_X16
#define VR_ATTN_TAIL_X16 0
#endif

namespace vr {

typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int HD> struct AttnCfg;
template <> struct AttnCfg<64>  { static constexpr int K32 = 2, TAIL = 0, DFRAGS = 4, PITCH = 160; };
template <> struct AttnCfg<72>  { static constexpr int K32 = 2, TAIL = 1, DFRAGS = 5, PITCH = 160; };
template <> struct AttnCfg<128> { static constexpr int K32 = 4, TAIL = 0, DFRAGS = 8, PITCH = 288; };

constexpr int ATT_KV = 64;          // keys per tile
constexpr float MAX_SLACK = 8.0f;   // log2 units the running max may lag behind before O is rescaled

__device__ __forceinline__ bf16x4 lds_tr_read(const char* p) {
    const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    return __builtin_bit_cast(bf16x4, r);
}

template <int HD, int QF>
__global__ __launch_bounds__(256, 2) void attention_kernel(AttnArgs p) {
    using C = AttnCfg<HD>;
    constexpr int K32 = C::K32, DFRAGS = C::DFRAGS, PITCH = C::PITCH;
    constexpr bool TAIL = C::TAIL != 0;
    constexpr int CPR = HD / 8;                   // 16-byte chunks per global row
    constexpr int QT = 64 * QF;                   // query rows per workgroup
    constexpr int NCH = (ATT_KV * CPR + 255) / 256;   // staging chunks per thread

    __shared__ __attribute__((aligned(16))) char smem[2 * ATT_KV * PITCH];
    char* Ks = smem;
    char* Vs = smem + ATT_KV * PITCH;

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

    // ---- zero the LDS row padding once (staging never overwrites it)
    for (int i = tid; i < (int)sizeof(smem) / 16; i += 256)
        reinterpret_cast<u32x4*>(smem)[i] = u32x4{0, 0, 0, 0};

    if constexpr (TAIL) {   // V column HD (= 72) := 1.0 in every key row: the PV MFMA then also yields the row sums
        __syncthreads();
        if (tid < ATT_KV) *reinterpret_cast<bf16_t*>(Vs + tid * PITCH + HD * 2) = (bf16_t)1.0f;
    }

    // ---- Q fragments (B operand of S^T): lane holds Q[q = fr][d = ks*32 + fq*8 .. +7]
    bf16x8 qf[QF][K32];
    bf16x4 qtail[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const int q = qs + (wave * QF + f) * 16 + fr;
        const bool ok = q < q_len;
#pragma unroll
        for (int ks = 0; ks < K32; ++ks) {
            u32x4 raw = {0, 0, 0, 0};
            if (ok) raw = *reinterpret_cast<const u32x4*>(qbase + (size_t)q * p.ldq + ks * 32 + fq * 8);
            qf[f][ks] = __builtin_bit_cast(bf16x8, raw);
        }
        u32x2 rt = {0, 0};
        if (TAIL && ok && K32 * 32 + fq * 4 < HD)
            rt = *reinterpret_cast<const u32x2*>(qbase + (size_t)q * p.ldq + K32 * 32 + fq * 4);
        qtail[f] = __builtin_bit_cast(bf16x4, rt);
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

    // Staging: every thread moves NCH 16-byte chunks of K and of V per tile.  Loads are
    // UNCONDITIONAL (row clamped to the last valid key): a load under a divergent branch makes
    // hipcc drain vmcnt(0) at the join and lose the prefetch.  Rows past kv_len are therefore
    // copies of the last key — harmless, their scores are masked to -inf (P = 0).
    u32x4 rk[NCH], rv[NCH];
    int st_key[NCH], st_ch[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = min(tid + i * 256, ATT_KV * CPR - 1);
        st_key[i] = c / CPR; st_ch[i] = c % CPR;
    }
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int kg = min(tile * ATT_KV + st_key[i], kv_len - 1);
            rk[i] = *reinterpret_cast<const u32x4*>(kbase + (size_t)kg * p.ldk + st_ch[i] * 8);
            rv[i] = *reinterpret_cast<const u32x4*>(vbase + (size_t)kg * p.ldv + st_ch[i] * 8);
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {   // (threads clamped to the last chunk rewrite the same bytes)
            *reinterpret_cast<u32x4*>(Ks + st_key[i] * PITCH + st_ch[i] * 16) = rk[i];
            *reinterpret_cast<u32x4*>(Vs + st_key[i] * PITCH + st_ch[i] * 16) = rv[i];
        }
    };

    // per-lane LDS offsets: K rows by fragment, V tr-read base (row fq*4 + fr/4, col-quad fr%4)
    const int k_off = fr * PITCH + fq * 16;
    const int v_off = (fq * 4 + (fr >> 2)) * PITCH + (fr & 3) * 8;

    load_tile(0);
    for (int tile = 0; tile < n_tiles; ++tile) {
        __syncthreads();            // previous tile fully consumed (and the zero-fill done)
        write_tile();
        __syncthreads();
        if (tile + 1 < n_tiles) load_tile(tile + 1);   // in flight during the MFMAs below

        const int key0 = tile * ATT_KV;
        // ---- S^T = K Q^T : all K fragments of the tile are fetched up front (each read once for
        //      all QF q-fragments), so the LDS latency is paid once and the MFMAs run back to back
        bf16x8 ka[4][K32];
        bf16x4 kt[4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            const char* kr = Ks + kf * 16 * PITCH + k_off;
#pragma unroll
            for (int ks = 0; ks < K32; ++ks) ka[kf][ks] = *reinterpret_cast<const bf16x8*>(kr + ks * 64);
            if constexpr (TAIL) kt[kf] = *reinterpret_cast<const bf16x4*>(Ks + (kf * 16 + fr) * PITCH + K32 * 64 + fq * 8);
        }
        f32x4 s[QF][4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
#pragma unroll
            for (int f = 0; f < QF; ++f) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < K32; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[kf][ks], qf[f][ks], acc, 0, 0, 0);
                if constexpr (TAIL) {
#if VR_ATTN_TAIL_X16
                    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kt[kf], qtail[f], acc, 0, 0, 0);
#else
                    const bf16x4 z = {};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_shufflevector(kt[kf], z, 0, 1, 2, 3, 4, 5, 6, 7),
                                                                  __builtin_shufflevector(qtail[f], z, 0, 1, 2, 3, 4, 5, 6, 7),
                                                                  acc, 0, 0, 0);
#endif
                }
                s[f][kf] = acc;
            }
        }
        // V^T fragments of the first PV k-step: issued now, consumed after the softmax below
        bf16x8 va0[DFRAGS];
#pragma unroll
        for (int d = 0; d < DFRAGS; ++d) {
            const char* vr = Vs + v_off + d * 32;
            va0[d] = __builtin_shufflevector(lds_tr_read(vr), lds_tr_read(vr + 16 * PITCH), 0, 1, 2, 3, 4, 5, 6, 7);
        }
        // ---- mask + online softmax; lane owns query q, keys key0 + kf*16 + fq*4 + r.
        // VALU diet (the kernel is VALU/MFMA balanced at head_dim 72): the scale is folded into one
        // fma per score, the running max is only raised when it grows by more than 2^MAX_SLACK
        // (so the O rescale is a rare wave-uniform branch; P stays <= 2^MAX_SLACK, harmless in
        // bf16/fp32), and for head_dim 72 the row sum comes out of the PV MFMA itself: column 72 of
        // the zero-padded V tile holds 1.0, so O^T[72][q] = sum_k P[k][q].
        const bool need_mask = (key0 + ATT_KV > kv_len) || (p.causal && key0 + ATT_KV > qs);
        bf16x8 pb[QF][2];
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            if (need_mask) {
                const int q = qs + (wave * QF + f) * 16 + fr;
                const int lim = p.causal ? min(kv_len - 1, q) : kv_len - 1;   // last visible key
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key0 + kf * 16 + fq * 4 + r > lim) s[f][kf][r] = -INFINITY;
            }
            float mx = fmaxf(fmaxf(s[f][0][0], s[f][0][1]), fmaxf(s[f][0][2], s[f][0][3]));
#pragma unroll
            for (int kf = 1; kf < 4; ++kf)
                mx = fmaxf(mx, fmaxf(fmaxf(s[f][kf][0], s[f][kf][1]), fmaxf(s[f][kf][2], s[f][kf][3])));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mxs = mx * sc;
            const bool upd = mxs > m_run[f] + MAX_SLACK;       // (-inf + slack = -inf: first valid tile updates)
            if (__any(upd)) {
                const float m_new = upd ? mxs : m_run[f];
                const float alpha = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(m_run[f] - m_new);
                m_run[f] = m_new;
                if constexpr (!TAIL) l_run[f] *= alpha;
#pragma unroll
                for (int d = 0; d < DFRAGS; ++d) o[f][d] *= alpha;
            }
            const float neg_m = (m_run[f] == -INFINITY) ? 0.f : -m_run[f];   // all-masked rows: exp2(-inf) = 0
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[f][kf][r] = __builtin_amdgcn_exp2f(fmaf(s[f][kf][r], sc, neg_m));
            if constexpr (!TAIL) {
                float rs = 0.f;
#pragma unroll
                for (int kf = 0; kf < 4; ++kf) rs += (s[f][kf][0] + s[f][kf][1]) + (s[f][kf][2] + s[f][kf][3]);
                rs += __shfl_xor(rs, 16, 64);
                rs += __shfl_xor(rs, 32, 64);
                l_run[f] += rs;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pb[f][ks][r] = f2bf(s[f][2 * ks][r]);
                    pb[f][ks][4 + r] = f2bf(s[f][2 * ks + 1][r]);
                }
        }
        // ---- O^T += V^T P^T : each V^T fragment (two transposing reads; rows (2ks)*16 + fq*4 + j
        //      and (2ks+1)*16 + fq*4 + j) feeds all QF q-fragments
        bf16x8 va1[DFRAGS];
#pragma unroll
        for (int d = 0; d < DFRAGS; ++d) {
            const char* vr = Vs + v_off + 32 * PITCH + d * 32;
            va1[d] = __builtin_shufflevector(lds_tr_read(vr), lds_tr_read(vr + 16 * PITCH), 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int d = 0; d < DFRAGS; ++d)
#pragma unroll
            for (int f = 0; f < QF; ++f)
                o[f][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va0[d], pb[f][0], o[f][d], 0, 0, 0);
#pragma unroll
        for (int d = 0; d < DFRAGS; ++d)
#pragma unroll
            for (int f = 0; f < QF; ++f)
                o[f][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va1[d], pb[f][1], o[f][d], 0, 0, 0);
    }

    // ---- normalise and store: lane owns out[q][h*HD + d*16 + fq*4 .. +3]
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const int q = qs + (wave * QF + f) * 16 + fr;
        float l = l_run[f];
        if constexpr (TAIL) l = __shfl(o[f][DFRAGS - 1][0], 32 + fr, 64);   // O^T[72][q]: lane (fq=2, fr=q), reg 0 (all lanes active here)
        if (q >= q_len) continue;
        const float inv = 1.0f / l;
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
