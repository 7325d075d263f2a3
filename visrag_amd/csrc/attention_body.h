// Flash-attention workgroup body shared by attention.hip (the grid form) and gen_persist.hip (the decode step's persistent
// kernel).  Design notes: attention.hip.
#pragma once
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace vr {

typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int HD> struct AttnCfg;
template <> struct AttnCfg<64>  { static constexpr int K32 = 2, TAIL = 0, DFRAGS = 4, PITCH = 160; };
template <> struct AttnCfg<72>  { static constexpr int K32 = 2, TAIL = 1, DFRAGS = 5, PITCH = 160; };
// head_dim 80 (the EVisRAG vision tower): the 72 form with every byte of the 160-byte row in use — the tail MFMA
// carries d = 64..79 (two of Q's four k-slots), and there is no spare V column for the row sums (VALU, like 64 / 128)
template <> struct AttnCfg<80>  { static constexpr int K32 = 2, TAIL = 1, DFRAGS = 5, PITCH = 160; };
template <> struct AttnCfg<128> { static constexpr int K32 = 4, TAIL = 0, DFRAGS = 8, PITCH = 288; };

#ifndef VR_ATTN_TAIL16
#define VR_ATTN_TAIL16 1
#endif
// the tail of head_dim 72 / 80 (d = 64..79) as ONE 16x16x16 MFMA per fragment — half the matrix-pipe time of the
// 16x16x32 form it replaces, whose second half multiplied zeros (72: 80 instead of 96 columns of QK^T work)
constexpr bool TAIL16 = VR_ATTN_TAIL16 != 0;
// wave priority: raised for a tile's PV MFMAs, dropped for its softmax (1) — with three workgroups per CU the arbiter then
// prefers the wave that can feed the matrix pipe and fills in with the others' VALU work: ViT attention 738 -> 764 TF in
// isolation, 5.28 -> 5.18 ms per step in the model.  (2: the other way round, 3 / 4: also raised for the score MFMAs, to
// level 3 / 1: 766-783 TF in isolation, no better than 1 in the model.)
#ifndef VR_ATTN_PRIO
#define VR_ATTN_PRIO 1
#endif
constexpr int ATT_KV = 64;          // keys per tile
constexpr float MAX_SLACK = 8.0f;   // log2 units the running max may lag behind before O is rescaled

__device__ __forceinline__ bf16x4 lds_tr_read(const char* p) {
    const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    return __builtin_bit_cast(bf16x4, r);
}

// reductions over the four lanes that hold the same query column (lane ^ 16, lane ^ 32)
__device__ __forceinline__ float col4_max(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float col4_sum(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// The kernel's body as a device function: `unit` = the (q tile, head, batch item) index a workgroup of the grid form takes
// from blockIdx, `smem` = attn_smem_bytes<HD, PIPE>() of LDS.  The decode step's persistent kernel (gen_persist.hip) calls it
// per (KV range, KV head) unit with q_lds = the group's query rows built in LDS (read instead of p.q) and COH = true: the
// partial rows and log-sum-exps leave as device-scope relaxed atomics (sc1), because workgroups on OTHER XCDs read them
// later in the same launch.
template <int HD, int PIPE> constexpr int attn_smem_bytes() { return 2 * (PIPE ? 2 : 1) * ATT_KV * AttnCfg<HD>::PITCH; }

template <int HD, int QF, int PIPE, bool COH = false>
__device__ __forceinline__ void attention_body(const AttnArgs& p, int unit, char* smem, const bf16_t* q_lds = nullptr) {
    using C = AttnCfg<HD>;
    constexpr int K32 = C::K32, DFRAGS = C::DFRAGS, PITCH = C::PITCH;
    constexpr bool TAIL = C::TAIL != 0;
    constexpr bool ONES = TAIL && DFRAGS * 16 > HD;    // a spare V column holds 1.0: the PV MFMA produces the row sums
    constexpr int TAILQ = (HD - K32 * 32) / 8;         // 16-byte chunks of real data in the tail window
    constexpr int CPR = HD / 8;                   // 16-byte chunks per global row
    constexpr int QT = 64 * QF;                   // query rows per workgroup
    constexpr int NCH = (ATT_KV * CPR + 255) / 256;   // staging chunks per thread
    constexpr int NB = PIPE ? 2 : 1;              // K / V slots
    constexpr bool DMA = PIPE >= 3;
    constexpr int SLOT = ATT_KV * PITCH;

    static_assert(2 * NB * SLOT == attn_smem_bytes<HD, PIPE>(), "LDS size");   // K slots, then V slots
    char* const Ks = smem;
    char* const Vs = smem + NB * SLOT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;

    const int q_tiles = (p.max_q + QT - 1) / QT;
    const int t = unit;
    const int qt = t % q_tiles, bh = t / q_tiles;
    const int h = bh % p.heads, b = bh / p.heads;

    const int kv0 = p.cu_kv[b], kv_len = (p.kv_end ? p.kv_end[b] : p.cu_kv[b + 1]) - kv0;
    const int q_row0 = p.cu_q[b];                      // rows of `out` (and of q unless shared)
    const int q_len = p.cu_q[b + 1] - q_row0;
    const int qs = qt * QT;                            // first query of this tile (seq-relative)
    if (qs >= q_len || kv_len <= 0) return;

    const bf16_t* qbase = q_lds ? q_lds : (const bf16_t*)p.q + (size_t)(p.q_in_rows ? p.q_in_rows[b] : (p.q_shared ? 0 : q_row0)) * p.ldq +
                                          h * (p.q_head_stride ? p.q_head_stride : HD);
    const int hkv = p.kv_group > 1 ? h / p.kv_group : h;          // grouped-query attention: K/V head of this query head
    const bf16_t* kbase = (const bf16_t*)p.k + (size_t)kv0 * p.ldk + hkv * HD;
    const bf16_t* vbase = (const bf16_t*)p.v + (size_t)kv0 * p.ldv + hkv * HD;
    // (wave-uniform: everything above derives from blockIdx and kernel arguments)
    const auto krsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, ((kv_len - 1) * p.ldk + HD) * 2, 0x00020000);
    const auto vrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, ((kv_len - 1) * p.ldv + HD) * 2, 0x00020000);

    // ---- zero the LDS (row padding stays zero: staging never overwrites it)
    for (int i = tid; i < attn_smem_bytes<HD, PIPE>() / 16; i += 256)
        reinterpret_cast<u32x4*>(smem)[i] = u32x4{0, 0, 0, 0};

    // ---- Q fragments (B operand of S^T): lane holds Q[q = fr][d = ks*32 + fq*8 .. +7]
    bf16x8 qf[QF][K32];
    bf16x8 qtail[QF];
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
        // tail MFMA (d = 64..95 window): lanes fq == 0 carry the real d = 64..71, every other k-slot of Q
        // is zero — so the K operand of those slots may be ANY finite LDS content (see scores())
        u32x4 rt = {0, 0, 0, 0};
        if constexpr (TAIL && TAIL16) {
            // 16x16x16 tail: lane (fr, fq) holds Q[q = fr][d = 64 + fq*4 .. +3] in the low half (d >= HD: zero)
            if (ok && 64 + fq * 4 < HD) {
                const u32x2 r2 = *reinterpret_cast<const u32x2*>(qbase + (size_t)q * p.ldq + K32 * 32 + fq * 4);
                rt[0] = r2[0]; rt[1] = r2[1];
            }
        } else {
            if (TAIL && ok && fq < TAILQ) rt = *reinterpret_cast<const u32x4*>(qbase + (size_t)q * p.ldq + K32 * 32 + fq * 8);
        }
        qtail[f] = __builtin_bit_cast(bf16x8, rt);
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

    // ---- staging through registers (PIPE 0..2): every thread moves NCH 16-byte chunks of K and of V
    //      per tile (threads past the last chunk repeat it: same bytes, same address)
    constexpr int NCHR = DMA ? 1 : NCH;
    u32x4 rk[NCHR], rv[NCHR];
    int st_lds[NCHR];
    unsigned st_k[NCHR], st_v[NCHR];
    if constexpr (!DMA) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = min(tid + i * 256, ATT_KV * CPR - 1);
            const int key = c / CPR, ch = c % CPR;
            st_lds[i] = key * PITCH + ch * 16;
            st_k[i] = (unsigned)(key * p.ldk + ch * 8) * 2u;
            st_v[i] = (unsigned)(key * p.ldv + ch * 8) * 2u;
        }
    }
    const unsigned k_step = (unsigned)(ATT_KV * p.ldk) * 2u, v_step = (unsigned)(ATT_KV * p.ldv) * 2u;
    auto load_k = [&](int tile) {
#pragma unroll
        for (int i = 0; i < NCHR; ++i) rk[i] = __builtin_amdgcn_raw_buffer_load_b128(krsrc, st_k[i] + tile * k_step, 0, 0);
    };
    auto load_v = [&](int tile) {
#pragma unroll
        for (int i = 0; i < NCHR; ++i) rv[i] = __builtin_amdgcn_raw_buffer_load_b128(vrsrc, st_v[i] + tile * v_step, 0, 0);
    };
    auto write_k = [&](int slot) {
#pragma unroll
        for (int i = 0; i < NCHR; ++i) *reinterpret_cast<u32x4*>(Ks + slot * SLOT + st_lds[i]) = rk[i];
    };
    auto write_v = [&](int slot) {
#pragma unroll
        for (int i = 0; i < NCHR; ++i) *reinterpret_cast<u32x4*>(Vs + slot * SLOT + st_lds[i]) = rv[i];
    };

    // ---- staging by LDS-DMA (PIPE 3).  A tile image is ATT_KV rows x CPL 16-byte chunks, lane-linear
    //      per wave instruction (64 chunks = 1 KiB: buffer_load_dwordx4 ... lds); the 2 * NDMA
    //      instructions of a K + V tile pair are dealt round-robin to the 4 waves.  Lanes that would
    //      land on a padding chunk are switched off (EXEC): the zero fill and the 1.0 column written
    //      once at kernel start stay in place.  Rows past kv_len are out of the descriptor's range and
    //      arrive as zeros.
    constexpr int CPL = PITCH / 16;
    constexpr int NDMA = ATT_KV * CPL / 64;
    constexpr int NI = DMA ? (2 * NDMA + 3) / 4 : 1;
    unsigned doff[NI];             // byte offset of this lane's chunk inside tile 0 (K or V)
    unsigned dmask = 0;            // bit i: this lane carries payload in instruction slot i
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    if constexpr (DMA) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = wave_u + 4 * i;                 // instruction index: K tile 0..NDMA-1, V tile NDMA..
            const int isv = j >= NDMA;
            const int c = (j - isv * NDMA) * 64 + lane;
            const int row = c / CPL, ch = c % CPL;
            doff[i] = (unsigned)(row * (isv ? p.ldv : p.ldk) + ch * 8) * 2u;
            if (ch < CPR && j < 2 * NDMA) dmask |= 1u << i;
        }
    }
    auto dma_issue = [&](int tile_k, int slot_k, int tile_v, int slot_v) {   // K tile -> K slot, V tile -> V slot
        if constexpr (DMA) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int j = wave_u + 4 * i;
                const int isv = j >= NDMA;
                char* dst = (isv ? Vs + slot_v * SLOT : Ks + slot_k * SLOT) + (j - isv * NDMA) * 1024;
                const unsigned off = doff[i] + (isv ? (unsigned)tile_v * v_step : (unsigned)tile_k * k_step);
                if (dmask & (1u << i)) {
                    if (isv) __builtin_amdgcn_raw_ptr_buffer_load_lds(vrsrc, VR_LDS(dst), 16, off, 0, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(krsrc, VR_LDS(dst), 16, off, 0, 0, 0);
                }
            }
        }
    };
    auto dma_tile = [&](int tile, int slot) { dma_issue(tile, slot, tile, slot); };

    // per-lane LDS offsets: K rows by fragment, V tr-read base (row fq*4 + fr/4, col-quad fr%4)
    const int k_off = fr * PITCH + fq * 16;
    const int v_off = (fq * 4 + (fr >> 2)) * PITCH + (fr & 3) * 8;

    // ---- S^T = K Q^T for one tile: K fragments fetched up front (each read once for all QF
    //      q-fragments); accumulators are walked ks-outermost so that dependent MFMAs are 8 apart
    auto scores = [&](const char* Kt, f32x4 (&s)[QF][4]) {
        bf16x8 ka[4][K32];
        bf16x8 kt[4];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            const char* kr = Kt + kf * 16 * PITCH + k_off;
#pragma unroll
            for (int ks = 0; ks < K32; ++ks) ka[kf][ks] = *reinterpret_cast<const bf16x8*>(kr + ks * 64);
            // tail: one b128 read per lane at d = 64 + fq*8.  fq 0: the real d 64..71; fq 1: the row's zero
            // padding; fq 2, 3: the first bytes of the NEXT row (finite K data, or the start of the next
            // slot / the V slots after the last row) — multiplied by Q's zero k-slots
            if constexpr (TAIL && TAIL16) {
                // d = 64 + fq*4 .. +3 of key row fr: 8 bytes at column byte 128 + fq*8 (72: d 72..79 is the row's zero padding)
                const u32x2 r2 = *reinterpret_cast<const u32x2*>(kr - fq * 16 + K32 * 64 + fq * 8);
                kt[kf] = __builtin_bit_cast(bf16x8, u32x4{r2[0], r2[1], 0u, 0u});
            } else if constexpr (TAIL) {
                kt[kf] = *reinterpret_cast<const bf16x8*>(kr + K32 * 64);
            }
        }
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int f = 0; f < QF; ++f) s[f][kf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < K32; ++ks)
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int f = 0; f < QF; ++f)
                    s[f][kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[kf][ks], qf[f][ks], s[f][kf], 0, 0, 0);
        if constexpr (TAIL && TAIL16) {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int f = 0; f < QF; ++f) {
                    const bf16x4 ka4 = __builtin_shufflevector(kt[kf], kt[kf], 0, 1, 2, 3);
                    const bf16x4 qb4 = __builtin_shufflevector(qtail[f], qtail[f], 0, 1, 2, 3);
                    s[f][kf] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, ka4), __builtin_bit_cast(s16x4, qb4),
                                                                         s[f][kf], 0, 0, 0);
                }
        } else if constexpr (TAIL) {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int f = 0; f < QF; ++f)
                    s[f][kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt[kf], qtail[f], s[f][kf], 0, 0, 0);
        }
    };

    // ---- online softmax of tile `tile` in three steps, so that the caller can put the next tile's
    // score MFMAs between the short statistics step and the long exp / convert step.
    // VALU diet (the kernel is VALU/MFMA balanced at head_dim 72): the scale is folded into one
    // fma per score, the running max is only raised when it grows by more than 2^MAX_SLACK (so the
    // O rescale is a rare wave-uniform branch; P stays <= 2^MAX_SLACK, harmless in bf16/fp32), and
    // for head_dim 72 the row sum comes out of the PV MFMA itself (V column 72 == 1.0).
    // 1. mask + running max (+ rare rescale of O); returns -m per q-fragment
    auto stats = [&](f32x4 (&s)[QF][4], int tile, float (&neg_m)[QF]) {
        const int key0 = tile * ATT_KV;
        const bool need_mask = (key0 + ATT_KV > kv_len) || (p.causal && key0 + ATT_KV > qs);
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
            mx = col4_max(mx);
            const float mxs = mx * sc;
            const bool upd = mxs > m_run[f] + MAX_SLACK;       // (-inf + slack = -inf: first valid tile updates)
            if (__any(upd)) {
                const float m_new = upd ? mxs : m_run[f];
                const float alpha = (m_new == -INFINITY) ? 1.0f : __builtin_amdgcn_exp2f(m_run[f] - m_new);
                m_run[f] = m_new;
                if constexpr (!ONES) l_run[f] *= alpha;
#pragma unroll
                for (int d = 0; d < DFRAGS; ++d) o[f][d] *= alpha;
            }
            neg_m[f] = (m_run[f] == -INFINITY) ? 0.f : -m_run[f];   // all-masked rows: exp2(-inf) = 0
        }
    };
    // 2. P = exp2(s * scale - m), packed to the bf16 B operand of the PV MFMA (branch-free)
    auto exp_pack = [&](f32x4 (&s)[QF][4], const float (&neg_m)[QF], bf16x8 (&pb)[QF][2]) {
#pragma unroll
        for (int f = 0; f < QF; ++f) {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[f][kf][r] = __builtin_amdgcn_exp2f(fmaf(s[f][kf][r], sc, neg_m[f]));
            if constexpr (!ONES) {
                float rs = 0.f;
#pragma unroll
                for (int kf = 0; kf < 4; ++kf) rs += (s[f][kf][0] + s[f][kf][1]) + (s[f][kf][2] + s[f][kf][3]);
                l_run[f] += col4_sum(rs);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pb[f][ks][r] = f2bf(s[f][2 * ks][r]);
                    pb[f][ks][4 + r] = f2bf(s[f][2 * ks + 1][r]);
                }
        }
    };
    // 3. O^T += V^T P^T: each V^T fragment (two transposing reads; rows (2ks)*16 + fq*4 + j and
    //    (2ks+1)*16 + fq*4 + j) feeds all QF q-fragments
    auto v_frags = [&](const char* Vt, int kstep, bf16x8 (&va)[DFRAGS]) {
        if constexpr (!DMA) {
#pragma unroll
            for (int d = 0; d < DFRAGS; ++d) {
                const char* vr = Vt + v_off + kstep * 32 * PITCH + d * 32;
                va[d] = __builtin_shufflevector(lds_tr_read(vr), lds_tr_read(vr + 16 * PITCH), 0, 1, 2, 3, 4, 5, 6, 7);
            }
        } else {
            // PIPE 3: the transposing reads go through inline asm.  hipcc cannot see that the tr-read
            // builtin does not alias the LDS-DMA of the OTHER slot and puts `s_waitcnt vmcnt(0)` in
            // front of the first one — which would wait for the tile that was requested moments ago.
            // Hidden in asm the reads are ours to count: v_wait() below (lgkmcnt(0), naming every
            // destination so that no consumer can be scheduled above it) closes them.
            const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(Vt) +
                                  (unsigned)(v_off + kstep * 32 * PITCH);
            u32x2 lo[DFRAGS], hi[DFRAGS];
#pragma unroll
            for (int d = 0; d < DFRAGS; ++d) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo[d]) : "v"(base), "i"(d * 32) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi[d]) : "v"(base), "i"(d * 32 + 16 * PITCH) : "memory");
            }
#pragma unroll
            for (int d = 0; d < DFRAGS; ++d)
                va[d] = __builtin_bit_cast(bf16x8, u32x4{lo[d][0], lo[d][1], hi[d][0], hi[d][1]});
        }
    };
    auto v_wait = [&](bf16x8 (&va)[DFRAGS]) {
        if constexpr (DMA) {
            static_assert(!DMA || DFRAGS == 4 || DFRAGS == 5, "operand list below");
            if constexpr (DFRAGS == 5)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]), "+v"(va[4]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]));
        }
    };
    auto pv = [&](const bf16x8 (&va)[DFRAGS], const bf16x8 (&pb)[QF][2], int kstep) {
#pragma unroll
        for (int d = 0; d < DFRAGS; ++d)
#pragma unroll
            for (int f = 0; f < QF; ++f)
                o[f][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[d], pb[f][kstep], o[f][d], 0, 0, 0);
    };
    auto softmax_pv = [&](f32x4 (&s)[QF][4], const char* Vt, int tile) {
        bf16x8 va0[DFRAGS], va1[DFRAGS], pb[QF][2];
        float neg_m[QF];
        v_frags(Vt, 0, va0);                     // issued now, consumed after the softmax
        if constexpr (VR_ATTN_PRIO == 1 || VR_ATTN_PRIO == 3 || VR_ATTN_PRIO == 4) __builtin_amdgcn_s_setprio(0);
        if constexpr (VR_ATTN_PRIO == 2) __builtin_amdgcn_s_setprio(2);
        stats(s, tile, neg_m);
        exp_pack(s, neg_m, pb);
        v_frags(Vt, 1, va1);
        v_wait(va0);
        if constexpr (VR_ATTN_PRIO == 1) __builtin_amdgcn_s_setprio(2);
        if constexpr (VR_ATTN_PRIO == 3) __builtin_amdgcn_s_setprio(3);
        if constexpr (VR_ATTN_PRIO == 4) __builtin_amdgcn_s_setprio(1);
        if constexpr (VR_ATTN_PRIO == 2) __builtin_amdgcn_s_setprio(0);
        pv(va0, pb, 0);
        v_wait(va1);
        pv(va1, pb, 1);
    };

    auto set_ones = [&]() {   // V column HD (= 72) := 1.0 in every key row of every slot
        if constexpr (ONES) {
            if (tid < NB * ATT_KV) *reinterpret_cast<bf16_t*>(Vs + tid * PITCH + HD * 2) = (bf16_t)1.0f;
        }
    };

    if constexpr (!DMA) { load_k(0); load_v(0); }
    __syncthreads();                // zero-fill done
    set_ones();

    if constexpr (PIPE == 0) {
        for (int tile = 0; tile < n_tiles; ++tile) {
            if (tile) __syncthreads();            // previous tile fully consumed
            write_k(0); write_v(0);
            __syncthreads();
            load_k(tile + 1); load_v(tile + 1);   // in flight during the MFMAs below (past the end: zeros)
            f32x4 s[QF][4];
            scores(Ks, s);
            softmax_pv(s, Vs, tile);
        }
    } else if constexpr (PIPE == 1) {
        write_k(0); write_v(0);
        load_k(1); load_v(1);
        __syncthreads();
        // (two iterations per trip so that the slot addresses are compile-time constants)
        auto body = [&](int tile, auto cur_c) {
            constexpr int cur = decltype(cur_c)::value;
            write_k(cur ^ 1); write_v(cur ^ 1);   // tile+1: its slot was last read before the previous barrier
            load_k(tile + 2); load_v(tile + 2);
            f32x4 s[QF][4];
            scores(Ks + cur * SLOT, s);
            softmax_pv(s, Vs + cur * SLOT, tile);
            __syncthreads();
        };
        for (int tile = 0; tile < n_tiles; tile += 2) {
            body(tile, std::integral_constant<int, 0>{});
            if (tile + 1 >= n_tiles) break;
            body(tile + 1, std::integral_constant<int, 1>{});
        }
    } else if constexpr (PIPE == 3) {
        // (dsrc points at tile 0 after the setup above; every dma_tile call moves it one tile on)
        dma_tile(0, 0);
        __syncthreads();                          // (the LDS-DMA in flight makes this wait vmcnt(0) too)
        auto body = [&](int tile, auto cur_c) {
            constexpr int cur = decltype(cur_c)::value;
            if (tile + 1 < n_tiles) dma_tile(tile + 1, cur ^ 1);   // its slot was last read before the previous barrier
            f32x4 s[QF][4];
            if constexpr (VR_ATTN_PRIO == 3) __builtin_amdgcn_s_setprio(3);
            if constexpr (VR_ATTN_PRIO == 4) __builtin_amdgcn_s_setprio(1);
            scores(Ks + cur * SLOT, s);
            softmax_pv(s, Vs + cur * SLOT, tile);
            __syncthreads();                      // tile+1 landed (vmcnt(0) + barrier), slot `cur` free
        };
        for (int tile = 0; tile < n_tiles; tile += 2) {
            body(tile, std::integral_constant<int, 0>{});
            if (tile + 1 >= n_tiles) break;
            body(tile + 1, std::integral_constant<int, 1>{});
        }
    } else if constexpr (PIPE == 4) {
        // LDS-DMA staging AND scores one tile ahead: at the top of iteration t the LDS holds K[t+1]
        // (slot (t+1)&1) and V[t] (slot t&1), s_cur = scores of tile t; the iteration requests K[t+2]
        // and V[t+1] (rows past the end arrive as zeros)
        dma_tile(0, 0);
        __syncthreads();
        dma_issue(1, 1, 0, 0);                    // K[1]; the V half re-fetches V[0] into its own slot (harmless)
        f32x4 s_a[QF][4], s_b[QF][4];
        scores(Ks, s_a);
        __syncthreads();
        auto body = [&](int tile, auto cur_c, f32x4 (&s_cur)[QF][4], f32x4 (&s_nxt)[QF][4]) {
            constexpr int cur = decltype(cur_c)::value;
            dma_issue(tile + 2, cur, tile + 1, cur ^ 1);
            const char* Vt = Vs + cur * SLOT;
            bf16x8 va0[DFRAGS], va1[DFRAGS], pb[QF][2];
            float neg_m[QF];
            v_frags(Vt, 0, va0);
            stats(s_cur, tile, neg_m);
            scores(Ks + (cur ^ 1) * SLOT, s_nxt);
            exp_pack(s_cur, neg_m, pb);
            v_frags(Vt, 1, va1);
            v_wait(va0);
            pv(va0, pb, 0);
            v_wait(va1);
            pv(va1, pb, 1);
            __syncthreads();
        };
        for (int tile = 0; tile < n_tiles; tile += 2) {
            body(tile, std::integral_constant<int, 0>{}, s_a, s_b);
            if (tile + 1 >= n_tiles) break;
            body(tile + 1, std::integral_constant<int, 1>{}, s_b, s_a);
        }
    } else {
        // invariant at the top of iteration t: LDS holds K[t+1] (slot (t+1)&1) and V[t] (slot t&1),
        // s_cur = scores of tile t, registers carry K[t+2] and V[t+1] (in flight)
        write_k(0); write_v(0);
        load_k(1);
        __syncthreads();
        f32x4 s_a[QF][4], s_b[QF][4];
        scores(Ks, s_a);
        write_k(1);
        load_k(2); load_v(1);
        __syncthreads();
        // two iterations per trip: the two score sets swap roles (no register copies) and the slot
        // addresses are compile-time constants
        auto body = [&](int tile, auto cur_c, f32x4 (&s_cur)[QF][4], f32x4 (&s_nxt)[QF][4]) {
            constexpr int cur = decltype(cur_c)::value;
            write_k(cur); write_v(cur ^ 1);
            load_k(tile + 3); load_v(tile + 2);
            const char* Vt = Vs + cur * SLOT;
            bf16x8 va0[DFRAGS], va1[DFRAGS], pb[QF][2];
            float neg_m[QF];
            v_frags(Vt, 0, va0);
            stats(s_cur, tile, neg_m);
            // one straight-line region: the score MFMAs of tile+1 (past the end: unused) run under
            // the exp / convert VALU work of tile `tile`
            scores(Ks + (cur ^ 1) * SLOT, s_nxt);
            exp_pack(s_cur, neg_m, pb);
            v_frags(Vt, 1, va1);
            pv(va0, pb, 0);
            pv(va1, pb, 1);
            __syncthreads();
        };
        for (int tile = 0; tile < n_tiles; tile += 2) {
            body(tile, std::integral_constant<int, 0>{}, s_a, s_b);
            if (tile + 1 >= n_tiles) break;
            body(tile + 1, std::integral_constant<int, 1>{}, s_b, s_a);
        }
    }

    // ---- normalise and store: lane owns out[q][h*HD + d*16 + fq*4 .. +3]
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const int q = qs + (wave * QF + f) * 16 + fr;
        float l = l_run[f];
        if constexpr (ONES) l = __shfl(o[f][DFRAGS - 1][0], 32 + fr, 64);   // O^T[72][q]: lane (fq=2, fr=q), reg 0 (all lanes active here)
        if (q >= q_len) continue;
        if (p.lse && fq == 0) {                                       // sum_k 2^(s_k * sc) = 2^m * l
            const float lv = m_run[f] + __log2f(l);
            if constexpr (COH) __hip_atomic_store(p.lse + (size_t)(q_row0 + q) * p.heads + h, lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else p.lse[(size_t)(q_row0 + q) * p.heads + h] = lv;
        }
        const float inv = 1.0f / l;
        bf16_t* orow = (bf16_t*)p.out + (size_t)(q_row0 + q) * p.ldo + h * HD;
#pragma unroll
        for (int d = 0; d < DFRAGS; ++d) {
            const int dd = d * 16 + fq * 4;
            if (dd < HD) {
                bf16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = f2bf(o[f][d][r] * inv);
                if constexpr (COH) {
                    const u32x2 w2 = __builtin_bit_cast(u32x2, ov);
                    __hip_atomic_store(reinterpret_cast<unsigned*>(orow + dd), w2[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(reinterpret_cast<unsigned*>(orow + dd) + 1, w2[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    *reinterpret_cast<bf16x4*>(orow + dd) = ov;
                }
            }
        }
    }
}


}  // namespace vr
