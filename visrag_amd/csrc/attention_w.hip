// SigLIP ViT self-attention (16 heads x 72, non-causal, N = h*w patches) with ONE WAVE PER SIMD and a hand-ordered
// instruction stream (F.scaled_dot_product_attention, vision_transformer.py:92-96).
//
// Why a second kernel: attention.hip's <72, 2, 3> form (32 query rows per wave, three workgroups per CU) keeps the matrix pipe
// 43 % busy — per tile a wave issues ~700 cycles of VALU (one v_exp_f32 = 4 slots) and ~700 cycles of MFMAs, and the counters
// show the two barely overlapping: the chain scores -> max -> exp -> pack -> PV is serial inside a wave and the
// three co-resident waves fall into step.  Here a wave owns its SIMD (4 waves = 256 query rows per workgroup, 64 per wave:
// half the LDS traffic per flop) and the overlap is written down instead of hoped for:
//
//   * the tile of 64 keys is processed as two HALVES of 32 keys; period h of the stream runs three independent things
//         scores(h+1)  24 MFMAs   S^T = K Q^T of the NEXT half      (matrix pipe)
//         softmax(h)   48 VALU    exp2 / pack of THIS half          (fills the MFMA gaps)
//         PV(h-1)      20 MFMAs   O^T += V^T P^T of the PREVIOUS half
//     one MFMA per slot, at most two or three other instructions behind it, `sched_barrier(0)` after every slot: hipcc keeps
//     the order it is given and only allocates registers.
//   * VALU diet.  Q is multiplied by scale * log2(e) once per workgroup (before its bf16 rounding in the prologue) and the score
//     MFMAs START from C = -m (the row's running maximum, an integer in the log2 domain): the accumulator comes out as
//     s - m and goes straight into v_exp_f32 — no fma per score.  The maximum is only CHECKED per period (15 v_max3 over the
//     lane's 32 values against one constant, no cross-lane traffic); the rare period in which some row's s - m exceeds 2^3
//     takes a block between two periods that raises m by an integer (so every rescale is an exact power of two: O, the
//     packed P not yet multiplied, and the scores already computed against the old m).
//   * the row sums come out of the PV MFMA: the d = 72 row of the V^T fragment is forced to 1.0 in registers (4 v_cndmask
//     per period) — the LDS image needs no constant column, so K / V tiles arrive by unmasked LDS-DMA (one 1 KiB
//     instruction per piece, five per wave and tile) into a ring of six tiles, three tiles ahead, behind ONE counted
//     `vmcnt` + barrier per tile.
//   * O^T (80 fp32 per lane) lives in hand-allocated accumulation registers (attention_w_acc.h; build.py checks that
//     hipcc never touches them); everything else (~210 registers) in the arch VGPRs, so no v_accvgpr traffic in the loop.
//
// Layouts (fragments, 160-byte row pitch, key permutation of P^T) are those of attention_body.h.
// Roofline: MFMA (4*N^2*D flop per head).
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace vr {

// O^T accumulator n = f * 5 + d (f = 16-query fragment of the wave, d = 16-column fragment of head_dim 80) lives in a[4n : 4n+3]
#define AW_FOR_EACH_ACC(X) \
    X(0, "a[0:3]", "a0", "a1", "a2", "a3") \
    X(1, "a[4:7]", "a4", "a5", "a6", "a7") \
    X(2, "a[8:11]", "a8", "a9", "a10", "a11") \
    X(3, "a[12:15]", "a12", "a13", "a14", "a15") \
    X(4, "a[16:19]", "a16", "a17", "a18", "a19") \
    X(5, "a[20:23]", "a20", "a21", "a22", "a23") \
    X(6, "a[24:27]", "a24", "a25", "a26", "a27") \
    X(7, "a[28:31]", "a28", "a29", "a30", "a31") \
    X(8, "a[32:35]", "a32", "a33", "a34", "a35") \
    X(9, "a[36:39]", "a36", "a37", "a38", "a39") \
    X(10, "a[40:43]", "a40", "a41", "a42", "a43") \
    X(11, "a[44:47]", "a44", "a45", "a46", "a47") \
    X(12, "a[48:51]", "a48", "a49", "a50", "a51") \
    X(13, "a[52:55]", "a52", "a53", "a54", "a55") \
    X(14, "a[56:59]", "a56", "a57", "a58", "a59") \
    X(15, "a[60:63]", "a60", "a61", "a62", "a63") \
    X(16, "a[64:67]", "a64", "a65", "a66", "a67") \
    X(17, "a[68:71]", "a68", "a69", "a70", "a71") \
    X(18, "a[72:75]", "a72", "a73", "a74", "a75") \
    X(19, "a[76:79]", "a76", "a77", "a78", "a79")

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int AW_HD = 72, AW_PITCH = 160, AW_KV = 64, AW_QT = 256, AW_DF = 5;
constexpr int AW_SLOT = AW_KV * AW_PITCH;          // 10 KiB: one K (or V) tile image
constexpr int AW_NSLOT = 6;                        // ring of six tiles: loads run three tiles ahead
constexpr int AW_VRING = AW_NSLOT * AW_SLOT;       // K ring, then V ring
constexpr int AW_SMEM = 2 * AW_NSLOT * AW_SLOT;    // 120 KiB: the ring
constexpr int AW_OPITCH = 144;                     // output staging: 256 rows x 144 bytes above the ring
constexpr int AW_SMEM_ALL = AW_SMEM + AW_QT * AW_OPITCH;   // 156 KiB
constexpr int AW_NPIECE = AW_SLOT / 1024;          // 10 LDS-DMA instructions per tile image
// timing diagnostics only (tagged builds; results are wrong): 1 no softmax VALU, 2 no maximum check, 4 no LDS reads, 8 no loads /
// barrier, 16 no PV MFMAs, 32 no score MFMAs, 64 no output stores, 128 no Q loads, 256 no start-up loads, 512 no start-up scores / raise
#ifndef AW_DBG
#define AW_DBG 0
#endif
constexpr float AW_SLACK = 8.0f;                   // log2 units a score may exceed the running maximum before the row is rescaled

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// The score MFMAs are asm as well: given builtin MFMAs hipcc allocates their results in the accumulation registers — on top of
// the hand-allocated O^T.  With "v" operands everything it allocates stays in the arch VGPRs.  Its hazard bookkeeping does
// not see into the statements: every consumer of a score accumulator sits >= 8 MFMAs behind the last MFMA that wrote it
// (the schedule in `period`), accumulate chains (D == C) need no wait states.
__device__ __forceinline__ void aw_mfma32_first(f32x4& d, const u32x4& a, const bf16x8& b, const f32x4& c) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
}
__device__ __forceinline__ void aw_mfma32_acc(f32x4& d, const u32x4& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void aw_mfma16_acc(f32x4& d, const u32x2& a, const s16x4& b) {
    asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}

// The softmax VALU work is asm for the ORDER alone: as plain expressions hipcc sinks all of it behind the period's last MFMA
// (pure operations are not held by sched_barrier across its IR-level code motion); volatile asm statements keep their
// program order.  No wait states needed: VALU -> VALU dependences are interlocked, and a v_exp_f32 result is first read
// (by its v_cvt_pk) several slots later.
__device__ __forceinline__ float aw_exp2(float x) { asm volatile("v_exp_f32 %0, %0" : "+v"(x)); return x; }
__device__ __forceinline__ unsigned aw_cvt_pk(float lo, float hi) {
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void aw_max3(float& m, float a, float b, float c) { asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c)); }
__device__ __forceinline__ void aw_max2(float& m, float a, float b) { asm volatile("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b)); }

// The two rare blocks between periods change state IN PLACE through tied asm operands: written as expressions every value
// they touch becomes a new SSA value, and hipcc pays for the join with ~50 register copies on the path that skips them.
__device__ __forceinline__ float aw_sub_inplace(float x, float d) { asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(d)); return x; }
__device__ __forceinline__ float aw_mask_inplace(float x, int key, int kv_len) {          // key >= kv_len -> -inf
    asm volatile("v_cmp_le_i32 vcc, %2, %1\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(x) : "v"(key), "s"(kv_len), "v"(-INFINITY) : "vcc");
    return x;
}
__device__ __forceinline__ unsigned aw_scale_pk_inplace(unsigned w, float alpha) {        // both bf16 halves times alpha (a power of two: exact)
    unsigned t;
    asm volatile("v_lshlrev_b32 %1, 16, %0\n\tv_and_b32 %0, 0xffff0000, %0\n\tv_mul_f32 %1, %1, %2\n\tv_mul_f32 %0, %0, %2\n\t"
                 "v_cvt_pk_bf16_f32 %0, %1, %0" : "+v"(w), "=&v"(t) : "v"(alpha));
    return w;
}

template <int N>
__device__ __forceinline__ void aw_pv_mfma(const bf16x8& va, const u32x4& pb) {
#define AW_X(n, R, C0, C1, C2, C3) \
    if constexpr (N == n) asm volatile("v_mfma_f32_16x16x32_bf16 " R ", %0, %1, " R : : "v"(va), "v"(pb) : C0, C1, C2, C3);
    AW_FOR_EACH_ACC(AW_X)
#undef AW_X
}
template <int N>
__device__ __forceinline__ void aw_acc_zero() {
#define AW_X(n, R, C0, C1, C2, C3) \
    if constexpr (N == n) asm volatile("v_accvgpr_write_b32 " C0 ", 0\n\tv_accvgpr_write_b32 " C1 ", 0\n\tv_accvgpr_write_b32 " C2 \
                                       ", 0\n\tv_accvgpr_write_b32 " C3 ", 0" : : : C0, C1, C2, C3);
    AW_FOR_EACH_ACC(AW_X)
#undef AW_X
}
template <int N>
__device__ __forceinline__ void aw_acc_read(f32x4& v) {
#define AW_X(n, R, C0, C1, C2, C3) \
    if constexpr (N == n) asm volatile("v_accvgpr_read_b32 %0, " C0 "\n\tv_accvgpr_read_b32 %1, " C1 "\n\tv_accvgpr_read_b32 %2, " C2 \
                                       "\n\tv_accvgpr_read_b32 %3, " C3 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
    AW_FOR_EACH_ACC(AW_X)
#undef AW_X
}
template <int N>
__device__ __forceinline__ void aw_acc_write(const f32x4& v) {
#define AW_X(n, R, C0, C1, C2, C3) \
    if constexpr (N == n) asm volatile("v_accvgpr_write_b32 " C0 ", %0\n\tv_accvgpr_write_b32 " C1 ", %1\n\tv_accvgpr_write_b32 " C2 \
                                       ", %2\n\tv_accvgpr_write_b32 " C3 ", %3" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : C0, C1, C2, C3);
    AW_FOR_EACH_ACC(AW_X)
#undef AW_X
}

__device__ __forceinline__ float aw_col4_max(float v) {       // over the four lanes that hold the same query column
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// LDS reads of the stream go through asm: hipcc neither waits for the LDS-DMA in flight in front of them nor counts them;
// every group is closed by aw_wait_* (lgkmcnt(0), naming each destination so that no consumer can move above it)
#define AW_DS_B128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define AW_DS_B64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define AW_DS_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))

}  // namespace

// NF = 16-query fragments per wave, NW = waves per workgroup (NF * NW = 16: 256 query rows):
//   <4, 4>  one wave per SIMD, the whole 512-register file per wave;
//   <2, 8>  two waves per SIMD — a single wave issues one instruction per ~5.3 clocks (measured: 2350 clocks per 64-key tile for
//           ~340 instructions around 88 MFMAs of 16 clocks each: issue-bound at 60 % of the matrix pipe), two waves with the same
//           hand-ordered stream feed the pipe alternately and hide each other's fillers.
template <int NF, int NW>
__device__ __forceinline__ void attention72w_body(const AttnArgs& p) {
    static_assert(NF * NW == 16 && (NF == 2 || NF == 4), "256 query rows per workgroup");
    constexpr int NS = 11 * NF;                     // MFMA slots per period: 4 NF score, 2 NF tail, 5 NF PV
    constexpr int S_TAIL = 4 * NF, S_PV = 6 * NF;
    constexpr int PW = (2 * AW_NPIECE + NW - 1) / NW;   // LDS-DMA instructions per wave and tile (NW = 8: 24 for 20 pieces, four repeated)
    constexpr int NV = 8 * NF;                      // scores per lane and half
    constexpr int S_CHECK = NF == 4 ? S_PV + 3 : S_PV + 2, S_KREAD = NF == 4 ? S_PV + 7 : S_PV + 3;
    static_assert(S_CHECK + NV / 2 <= NS && S_KREAD + 6 <= NS, "schedule fits");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    const int q_tiles = (p.max_q + AW_QT - 1) / AW_QT;
    const int total = p.B * p.heads * q_tiles;
    // PERSISTENT: workgroup w of the grid walks units w, w + grid, ... (the four query tiles of a head run at the same time on
    // neighbouring workgroups of one XCD and share its K / V in that L2).  A workgroup alone on its CU pays every latency of a
    // unit's start and end in full — measured per unit of 16 tiles (18 us of loop): Q loads 2.4 us, the first K / V tiles 2.0,
    // output stores 1.5, launch and the rest 3.9 — so the NEXT unit's K / V tiles are requested by the loader of this one (its
    // tiles past the end used to be out-of-range dummies), its Q rows are requested when the last score MFMAs have issued, and
    // the stores are left in flight.
    const int nwg = gridDim.x;
    const int lw = xcd_remap(blockIdx.x, nwg);
    const float sc = p.scale * 1.44269504088896340736f;
    const unsigned tile_step = (unsigned)(AW_KV * p.ldk) * 2u;
    // ONE descriptor per unit for K and V (launcher: v >= k, ldk == ldv, everything below 2 GiB): rows past kv_len read the next
    // sequence's rows or, past the end, zeros — finite either way, and their scores are masked / their P is exactly 0
    const unsigned vdelta = (unsigned)((const char*)p.v - (const char*)p.k);

    struct Unit { int u, ok, h, qs, q_len, kv_len, q_row0, kv0, T; };
    auto get_unit = [&](int u) -> Unit {
        Unit x{u, 0, 0, 0, 0, 0, 0, 0, 0};
        if (u >= total) return x;
        const int qt = u % q_tiles, bh = u / q_tiles, b = bh / p.heads;
        x.h = bh % p.heads;
        x.kv0 = p.cu_kv[b]; x.kv_len = p.cu_kv[b + 1] - x.kv0;
        x.q_row0 = p.cu_q[b]; x.q_len = p.cu_q[b + 1] - x.q_row0;
        x.qs = qt * AW_QT;
        x.T = (x.kv_len + AW_KV - 1) / AW_KV;
        x.ok = x.qs < x.q_len && x.kv_len > 0;
        return x;
    };
    auto next_valid = [&](int u) -> Unit {             // the first unit >= u of this workgroup's walk that has rows
        Unit x = get_unit(u);
        while (!x.ok && u < total) { u += nwg; x = get_unit(u); }
        return x;
    };
    auto unit_rsrc = [&](const Unit& x) {
        const bf16_t* kb = (const bf16_t*)p.k + (size_t)x.kv0 * p.ldk + x.h * AW_HD;
        return __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, x.ok ? vdelta + ((x.kv_len - 1) * p.ldv + AW_HD) * 2 : 0, 0x00020000);
    };

    Unit cur = next_valid(lw);
    if (!cur.ok) return;

    // ---- LDS-DMA pieces of this wave: instruction j = wave + NW i of the 20 of a tile pair (K 0..9, V 10..19; j >= 20 repeats
    //      piece j - 20: every wave issues PW per tile, so the counted waits hold); a piece is lane-linear: chunk
    //      c = piece * 64 + lane -> row c / 10, 16-byte chunk c % 10 (chunk 9 = the row's padding: whatever follows the head's
    //      144 bytes in memory — finite).  doff = the lane's byte offset in the LOADER's unit of the next tile to request.
    unsigned doff[PW];
    int dbase[PW];
    auto doff_reset = [&]() {
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            int j = wave + NW * i;
            if (j >= 2 * AW_NPIECE) j -= 2 * AW_NPIECE;
            const int isv = j >= AW_NPIECE;
            const int c = (j - isv * AW_NPIECE) * 64 + lane;
            const int row = c / 10, ch = c % 10;
            doff[i] = (unsigned)(row * p.ldk + ch * 8) * 2u + (isv ? vdelta : 0u);
            dbase[i] = __builtin_amdgcn_readfirstlane(isv * AW_VRING + (j - isv * AW_NPIECE) * 1024);
        }
    };
    auto ldrsrc = unit_rsrc(cur);                        // the loader's descriptor
    auto dma_piece = [&](int i, int slot_off) {          // piece i of the next tile to load -> ring slot at byte slot_off
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ldrsrc, VR_LDS(smem + dbase[i] + slot_off), 16, doff[i], 0, 0, 0);
        doff[i] += tile_step;
    };
    constexpr int P_LEAD = (PW + 1) / 2;                 // pieces of a tile requested under the odd period behind the barrier; the rest under the next even period
    int slot_ld = 0;                                     // ring slot (bytes) of the tile whose pieces are being requested

    // ---- Q rows of a unit: raw loads (requested early), then the multiplication by scale * log2(e) before the bf16 rounding
    //      (B operand of S^T: lane holds Q[q = fr][d = ks*32 + fq*8 .. +7]; tail: Q[q][64 + fq*4 .. +3], zero for d >= 72)
    u32x4 qraw[NF][2];
    u32x2 qtraw[NF];
    auto q_load = [&](const Unit& x) {
        const bf16_t* qbase = (const bf16_t*)p.q + (size_t)x.q_row0 * p.ldq + x.h * AW_HD;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int q = x.qs + (wave * NF + f) * 16 + fr;
            const bool ok = x.ok && q < x.q_len && !(AW_DBG & 128);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qraw[f][ks] = u32x4{0, 0, 0, 0};
                if (ok) qraw[f][ks] = *reinterpret_cast<const u32x4*>(qbase + (size_t)q * p.ldq + ks * 32 + fq * 8);
            }
            qtraw[f] = u32x2{0, 0};
            if (ok && 64 + fq * 4 < AW_HD) qtraw[f] = *reinterpret_cast<const u32x2*>(qbase + (size_t)q * p.ldq + 64 + fq * 4);
        }
    };
    bf16x8 qf[NF][2];
    s16x4 qtail[NF];
    auto q_scale = [&]() {
        if (p.q_prescaled) {              // (wave-uniform) the qkv GEMM's epilogue has multiplied them before ITS rounding
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                qf[f][0] = __builtin_bit_cast(bf16x8, qraw[f][0]); qf[f][1] = __builtin_bit_cast(bf16x8, qraw[f][1]);
                qtail[f] = __builtin_bit_cast(s16x4, qtraw[f]);
            }
            return;
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 v = __builtin_bit_cast(bf16x8, qraw[f][ks]);
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[f][ks][e] = f2bf(bf2f(v[e]) * sc);
            }
            const bf16x4 t4 = __builtin_bit_cast(bf16x4, qtraw[f]);
            bf16x4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = f2bf(bf2f(t4[e]) * sc);
            qtail[f] = __builtin_bit_cast(s16x4, t);
        }
    };

    f32x4 negm4[NF];                    // -m of the wave's query fragments (all four components equal): C of the score MFMAs
    f32x4 sA[NF][2], sB[NF][2];         // scores of two halves: [q fragment][16-key fragment of the half]
    u32x4 pbA[NF], pbB[NF];             // packed P^T of two halves (B operand of PV): word c = bf16 elements 2c, 2c + 1

    // per-lane LDS addresses (ring slot added per tile): K fragment rows, K tail, V transposing read
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)VR_LDS(smem);
    const unsigned k_lane = lds0 + (unsigned)(fr * AW_PITCH + fq * 16);
    const unsigned kt_lane = lds0 + (unsigned)(fr * AW_PITCH + 128 + fq * 8);
    const unsigned v_lane = lds0 + (unsigned)(AW_VRING + (fq * 4 + (fr >> 2)) * AW_PITCH + (fr & 3) * 8);
    const bool row8 = fr == 8;          // the V^T fragment row d = 72: forced to 1.0 (row sums out of the PV MFMA)
    int kv_len = cur.kv_len;            // of the unit being computed (mask_half)

#ifdef AW_TIMING
    // tile anatomy (tools/attn_anatomy.py; tagged builds, p.lse = u64 [unit][wave][64]): 0 start, 1 loop entry, 2 loop exit, 3 end,
    // 4 sum of (barrier release - arrival at the wait), 5 number of barriers, 8.. the first 24 barrier release stamps
    unsigned long long tm_start = __builtin_amdgcn_s_memtime(), tm_wait = 0, tm_loop0 = 0, tm_loop1 = 0;
    unsigned long long* tmrec = (unsigned long long*)p.lse + ((size_t)cur.u * NW + wave) * 64;
    int tm_nb = 0;
#endif
    u32x4 ka[2][2];                     // K fragments of the half to be scored next: [16-key fragment][k-step]
    u32x2 kt[2];
    u32x2 vlo[AW_DF], vhi[AW_DF];       // V^T fragments of the half in PV
    float mxall = 0.f, mxb = 0.f;       // max of the lane's scores of the half just computed (relative to m); mxb: the second chain

    // K fragments of half `hf` (0 / 1) of the tile in ring slot `ka_addr` / `kta_addr`
    // (generic lambdas: clang decides the captures of names used only with template-dependent indices too late — name them once)
    auto k_read = [&](auto u_c, auto hf_c, unsigned ka_addr, unsigned kta_addr) {
        (void)&ka; (void)&kt;
        constexpr int u = decltype(u_c)::value, hf = decltype(hf_c)::value;      // u = 0..5: (fragment, piece)
        constexpr int kf = u / 3, pc = u % 3, row = (hf * 2 + kf) * 16 * AW_PITCH;
        if constexpr (pc == 0) AW_DS_B128(ka[kf][0], ka_addr, row);
        else if constexpr (pc == 1) AW_DS_B128(ka[kf][1], ka_addr, row + 64);
        else AW_DS_B64(kt[kf], kta_addr, row);
    };
    auto k_wait = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ka[0][0]), "+v"(ka[0][1]), "+v"(ka[1][0]), "+v"(ka[1][1]), "+v"(kt[0]), "+v"(kt[1]));
    };
    auto v_read = [&](auto u_c, auto hf_c, unsigned v_addr) {
        (void)&vlo; (void)&vhi;
        constexpr int u = decltype(u_c)::value, hf = decltype(hf_c)::value;      // u = 0..9: (d fragment, lo / hi)
        constexpr int d = u / 2, off = hf * 32 * AW_PITCH + d * 32;
        if constexpr ((u & 1) == 0) AW_DS_TR(vlo[d], v_addr, off);
        else AW_DS_TR(vhi[d], v_addr, off + 16 * AW_PITCH);
    };
    auto v_wait = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[0]), "+v"(vhi[0]), "+v"(vlo[1]), "+v"(vhi[1]), "+v"(vlo[2]), "+v"(vhi[2]),
                     "+v"(vlo[3]), "+v"(vhi[3]), "+v"(vlo[4]), "+v"(vhi[4]));
    };

    // keys past kv_len of half `hh` (global half index) -> -inf, and the lane's maximum again
    auto mask_half = [&](f32x4 (&s)[NF][2], int hh) {
        float mx = -INFINITY;
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[f][kf][r] = aw_mask_inplace(s[f][kf][r], hh * 32 + kf * 16 + fq * 4 + r, kv_len);
                    mx = fmaxf(mx, s[f][kf][r]);
                }
        mxall = mx;
    };

    // The block between two periods that raises the running maxima (rare after the first tiles; `force`: the first half,
    // whose scores were computed against m = 0).  A column's m goes up by ceil(its largest s - m): every factor is an exact
    // power of two.  Scaled exactly once: O (all PV MFMAs issued so far have retired behind the wait states), the packed P of
    // the half whose PV has NOT been issued yet (`pbn`), and the scores already computed against the old m (`s`).
    auto raise = [&](f32x4 (&s)[NF][2], u32x4 (&pbn)[NF], auto force_c) {
        constexpr bool force = decltype(force_c)::value;      // the unit's first half: O and P are still zero, only m and the scores move
        if constexpr (!force) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            float mx = fmaxf(fmaxf(fmaxf(s[f][0][0], s[f][0][1]), fmaxf(s[f][0][2], s[f][0][3])),
                             fmaxf(fmaxf(s[f][1][0], s[f][1][1]), fmaxf(s[f][1][2], s[f][1][3])));
            mx = aw_col4_max(mx);
            const bool upd = (force && mx != -INFINITY) || mx > AW_SLACK;
            const float delta = upd ? __builtin_ceilf(mx) : 0.f;
            if (!__any(delta != 0.f)) continue;
            [[maybe_unused]] const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
            for (int e = 0; e < 4; ++e) negm4[f][e] = aw_sub_inplace(negm4[f][e], delta);
#pragma unroll
            for (int kf = 0; kf < 2; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[f][kf][r] = aw_sub_inplace(s[f][kf][r], delta);
            if constexpr (!force) {
#pragma unroll
            for (int e = 0; e < 4; ++e) pbn[f][e] = aw_scale_pk_inplace(pbn[f][e], alpha);
            static_for<0, AW_DF>([&](auto d_c) {
                constexpr int d = decltype(d_c)::value;
                // (f is a loop variable of an unrolled loop: select the accumulator by a constant switch)
                f32x4 o;
#define AW_RW(F) if constexpr (F < NF) { if (f == F) { aw_acc_read<F * AW_DF + d>(o); o *= alpha; aw_acc_write<F * AW_DF + d>(o); } }
                AW_RW(0) AW_RW(1) AW_RW(2) AW_RW(3)
#undef AW_RW
            });
            }
        }
        asm volatile("s_nop 3" ::: "memory");
        // the lane's maximum against the new m (only this half's scores moved)
        float mx = -INFINITY;
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int kf = 0; kf < 2; ++kf) mx = fmaxf(mx, fmaxf(fmaxf(s[f][kf][0], s[f][kf][1]), fmaxf(s[f][kf][2], s[f][kf][3])));
        mxall = mx;
    };

    // ---- one period.  SC: score MFMAs of half hn = h + 1 into s_nxt (K fragments in registers) + the K reads of half h + 2;
    //      SM: softmax of s_cur -> pb_new;  PV: V reads + PV MFMAs of half h - 1 (pb_old);  ODD: the tile's barrier;  LD: loads.
    //      ka_addr / kta_addr: K ring slot of the half to READ (h + 2), v_addr: V ring slot of half h - 1.
    auto period = [&](auto sc_c, auto sm_c, auto pv_c, auto odd_c, auto ld_c, f32x4 (&s_cur)[NF][2], f32x4 (&s_nxt)[NF][2], u32x4 (&pb_new)[NF],
                      u32x4 (&pb_old)[NF], unsigned ka_addr, unsigned kta_addr, unsigned v_addr, int hn) {
        constexpr bool SC = decltype(sc_c)::value, SM = decltype(sm_c)::value, PV = decltype(pv_c)::value, ODD = decltype(odd_c)::value;
        constexpr bool LD = decltype(ld_c)::value;    // the period carries its share of the loader's tile
        (void)&ka; (void)&kt; (void)&qf; (void)&qtail; (void)&negm4; (void)&vlo; (void)&vhi; (void)&doff; (void)&dbase; (void)&mxall; (void)&mxb; (void)&slot_ld;
        constexpr int PAR = ODD ? 1 : 0;
        if constexpr (ODD) {
            // tile (this + 2) has landed in every wave's share (the two younger tiles stay in flight), nobody reads the
            // slot of tile (this - 1) any more: its V half 1 went into registers in the even period
#ifdef AW_TIMING
            const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
            if constexpr (!(AW_DBG & 8)) {
                if constexpr (PW == 5) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
            }
#ifdef AW_TIMING
            const unsigned long long tw1 = __builtin_amdgcn_s_memtime();
            tm_wait += tw1 - tw0;
            if (lane == 0 && tm_nb < 24) tmrec[8 + tm_nb] = tw1;
            ++tm_nb;
#endif
        }
        if constexpr (SC && !(AW_DBG & 4)) k_wait();
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 va[AW_DF];
        static_for<0, NS>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            // ---- the slot's MFMA
            if constexpr (SC && i < S_TAIL) {
                constexpr int ks = i / (2 * NF), kf = (i / NF) % 2, f = i % NF;
                if constexpr (AW_DBG & 32) {}
                else if constexpr (ks == 0) aw_mfma32_first(s_nxt[f][kf], ka[kf][0], qf[f][0], negm4[f]);
                else aw_mfma32_acc(s_nxt[f][kf], ka[kf][1], qf[f][1]);
            } else if constexpr (SC && i < S_PV) {
                constexpr int kf = (i - S_TAIL) / NF, f = i % NF;
                if constexpr (!(AW_DBG & 32)) aw_mfma16_acc(s_nxt[f][kf], kt[kf], qtail[f]);
            } else if constexpr (PV && i >= S_PV) {
                constexpr int d = (i - S_PV) / NF, f = i % NF;
                if constexpr (!(AW_DBG & 16)) aw_pv_mfma<f * AW_DF + d>(va[d], pb_old[f]);
            }
            // ---- V^T fragments of half h - 1: requested under the first score MFMAs, complete before the first PV MFMA
            if constexpr (PV && i < 10 && !(AW_DBG & 4)) v_read(i_c, std::integral_constant<int, 1 - PAR>{}, v_addr);
            if constexpr (PV && i == S_PV - 2) {
                if constexpr (!(AW_DBG & 4)) v_wait();
#pragma unroll
                for (int d = 0; d < AW_DF; ++d) va[d] = __builtin_bit_cast(bf16x8, u32x4{vlo[d][0], vlo[d][1], vhi[d][0], vhi[d][1]});
            }
            if constexpr (PV && i == S_PV - 1) {
                u32x4 t = __builtin_bit_cast(u32x4, va[AW_DF - 1]);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = row8 ? 0x3F803F80u : t[e];
                va[AW_DF - 1] = __builtin_bit_cast(bf16x8, t);
            }
            // ---- softmax of s_cur: 12 NF units — exp f0 x8, then per further fragment (exp f x8, pack f-1 x4), last pack x4; one
            //      unit beside each V read, two per slot behind them
            if constexpr (SM && !(AW_DBG & 1)) {
                constexpr int u0 = i < 10 ? i : 10 + 2 * (i - 10);
                constexpr int un = i < 10 ? 1 : 2;
                static_for<u0, (u0 + un < 12 * NF ? u0 + un : 12 * NF)>([&](auto u_c) {
                    constexpr int u = decltype(u_c)::value;
                    // blocks: [0, 8) exp 0; then for g >= 1: [8 + 12 (g - 1), +8) exp g, [+8, +12) pack g - 1; the last 4: pack NF - 1
                    constexpr bool tail_pack = u >= 12 * NF - 4;
                    constexpr int g = u < 8 ? 0 : (u - 8) / 12 + 1, w = u < 8 ? u : (u - 8) % 12;
                    if constexpr (!tail_pack && (u < 8 || w < 8)) {
                        s_cur[g][w / 4][w % 4] = aw_exp2(s_cur[g][w / 4][w % 4]);
                    } else {
                        constexpr int f = tail_pack ? NF - 1 : g - 1;
                        constexpr int c = tail_pack ? u - (12 * NF - 4) : w - 8;
                        constexpr int kf = c / 2, r0 = (c % 2) * 2;
                        pb_new[f][c] = aw_cvt_pk(s_cur[f][kf][r0], s_cur[f][kf][r0 + 1]);      // word c = elements 2c, 2c + 1 = (kf * 4 + r0, + 1)
                    }
                });
            }
            // ---- the lane's maximum over the NV new scores, in the order the tail MFMAs finished them (value w: key fragment
            //      w / (4 NF), query fragment (w / 4) % NF, register w % 4): NV / 2 units from slot S_CHECK
            if constexpr (SC && i >= S_CHECK && i < S_CHECK + NV / 2 && !(AW_DBG & 2)) {
                constexpr int c = i - S_CHECK;
                auto val = [&](auto w_c) -> float { constexpr int w = decltype(w_c)::value; return s_nxt[(w / 4) % NF][w / (4 * NF)][w % 4]; };
                // (two chains: a unit that reads the result of the asm statement right in front of it gets an `s_nop` from hipcc)
                if constexpr (c == 0)
                    aw_max3(mxall, val(std::integral_constant<int, 0>{}), val(std::integral_constant<int, 1>{}), val(std::integral_constant<int, 2>{}));
                else if constexpr (c == 1)
                    aw_max3(mxb, val(std::integral_constant<int, 3>{}), val(std::integral_constant<int, 4>{}), val(std::integral_constant<int, NV - 1>{}));
                else if constexpr (c < NV / 2 - 1) {
                    if constexpr (c % 2 == 0) aw_max3(mxall, mxall, val(std::integral_constant<int, 2 * c + 1>{}), val(std::integral_constant<int, 2 * c + 2>{}));
                    else aw_max3(mxb, mxb, val(std::integral_constant<int, 2 * c + 1>{}), val(std::integral_constant<int, 2 * c + 2>{}));
                } else
                    aw_max2(mxall, mxall, mxb);
            }
            // (gfx950's K = 32 MFMAs read the second half of their A / B operands passes after they issue: a VALU result
            // allocated into a just-"dead" operand register two instructions behind the MFMA corrupts it — measured: the
            // row sums of fragment 0 off by 5 % in 1 % of the rows.  hipcc does not know the statements are MFMAs, so the
            // operands are kept alive by hand: K fragments until two slots behind the last score MFMA, V^T fragments and the
            // packed P until the end of the period.)
            if constexpr (SC && i == S_KREAD - 1)
                asm volatile("" : : "v"(ka[0][0]), "v"(ka[0][1]), "v"(ka[1][0]), "v"(ka[1][1]), "v"(kt[0]), "v"(kt[1]));
            // ---- K fragments of half h + 2 (the registers are free: the score MFMAs of this period have issued)
            if constexpr (SC && i >= S_KREAD && i < S_KREAD + 6 && !(AW_DBG & 4))
                k_read(std::integral_constant<int, i - S_KREAD>{}, std::integral_constant<int, PAR>{}, ka_addr, kta_addr);
            // ---- the loads of tile (this + 5) into the slot of tile (this - 1): the first (PW + 1) / 2 pieces under the odd
            //      period that follows the tile's barrier (after which nobody reads that slot), the rest under the next even
            //      period — PW between two barriers, spread over the slots
            if constexpr (LD && !(AW_DBG & 8)) {
                constexpr int n_here = ODD ? P_LEAD : PW - P_LEAD, first = ODD ? 0 : P_LEAD;
                static_for<0, n_here>([&](auto k_c) {
                    constexpr int k = decltype(k_c)::value;
                    if constexpr (i == NS * (k + 1) / (n_here + 1)) dma_piece(first + k, slot_ld);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (PV) {
#pragma unroll
            for (int d = 0; d < AW_DF; ++d) asm volatile("" : : "v"(va[d]));
#pragma unroll
            for (int f = 0; f < NF; ++f) asm volatile("" : : "v"(pb_old[f]));
        }
        // ---- between two periods: mask the keys past the end, raise the maxima if a score ran away (both rare)
        if constexpr (SC) {
            if (__builtin_expect(hn * 32 + 32 > kv_len, 0)) mask_half(s_nxt, hn);
            if (__builtin_expect(__any(mxall > AW_SLACK), 0)) raise(s_nxt, pb_new, std::false_type{});
        }
    };

    using T_ = std::true_type;
    using F_ = std::false_type;

    // ring slots (bytes): tile t - 1, t, t + 1 of the unit being computed
    int sl_prev = (AW_NSLOT - 1) * AW_SLOT, sl_cur = 0, sl_next = AW_SLOT;
    auto advance = [&]() {
        sl_prev = sl_cur; sl_cur = sl_next;
        sl_next = sl_next + AW_SLOT == AW_NSLOT * AW_SLOT ? 0 : sl_next + AW_SLOT;
    };
    auto advance_ld = [&]() { slot_ld = slot_ld + AW_SLOT == AW_NSLOT * AW_SLOT ? 0 : slot_ld + AW_SLOT; };
    // (the loader is five tiles ahead: tile t + 5 goes to the slot of tile t - 1, its first pieces under the odd period of tile t —
    // behind its barrier, after which nobody reads that slot — the rest under the even period of tile t + 1; past the unit's last
    // tile it requests the NEXT unit's first tiles)

    bool cold = true;                    // nothing of `cur` has been requested yet
    while (true) {
        const Unit nxt = next_valid(cur.u + nwg);
        // this unit's tail requests the next one's start when both are long enough for the loader to change units inside the main
        // loop (T >= 6: the ViT shapes); otherwise the queue is drained and the next unit starts cold
        const bool chain = nxt.ok && cur.T >= 6 && nxt.T >= 6;
        const int T = cur.T;
        kv_len = cur.kv_len;
        if (cold) {
            // ---- tiles 0 .. 3 and the leading pieces of tile 4 requested (rows past the end: zeros, no traffic — every tile is PW
            //      instructions per wave, so the counted waits hold), then the Q rows
            ldrsrc = unit_rsrc(cur);
            doff_reset();
            slot_ld = 0; sl_prev = (AW_NSLOT - 1) * AW_SLOT; sl_cur = 0; sl_next = AW_SLOT;
            if (!(AW_DBG & 256)) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
#pragma unroll
                    for (int i = 0; i < PW; ++i) dma_piece(i, slot_ld);
                    slot_ld += AW_SLOT;
                }
#pragma unroll
                for (int i = 0; i < P_LEAD; ++i) dma_piece(i, slot_ld);
            } else {
                slot_ld = 4 * AW_SLOT;
            }
            q_load(cur);
            q_scale();
        }
        static_for<0, NF * AW_DF>([&](auto n_c) { aw_acc_zero<decltype(n_c)::value>(); });
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            negm4[f] = f32x4{0.f, 0.f, 0.f, 0.f};
            pbA[f] = u32x4{0, 0, 0, 0};
            pbB[f] = pbA[f];
            // The values become opaque HERE: left as known zeros, hipcc materialises the C operand of the first score MFMAs in the
            // instruction in front of them (measured: `v_mov_b64 v[52:53], 0` then `v_mfma ..., v[52:55]`) — a VALU write the MFMA
            // needs two wait states behind, which hipcc does not pad for an asm statement: fragment 0 started from a stale C in ~1 % of
            // the workgroups.  build.py checks the listing for this pattern (mfma_operand_hazards).
            asm volatile("" : "+v"(negm4[f]), "+v"(pbA[f]), "+v"(pbB[f]));
        }

        // ---- start: tiles 0 and 1 visible (younger: tiles 2, 3 and the leading pieces of 4 — 2 PW + P_LEAD; Q loads and the
        //      previous unit's stores in the queue only make the wait longer: loads complete in order), scores of half 0
        //      against m = 0, then m
        if (!(AW_DBG & 256)) {
            if constexpr (PW == 5) asm volatile("s_waitcnt vmcnt(13)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        }
        static_for<0, 6>([&](auto u_c) { k_read(u_c, std::integral_constant<int, 0>{}, k_lane + sl_cur, kt_lane + sl_cur); });
        k_wait();
        static_for<0, S_PV>([&](auto i_c) {
            constexpr int i = decltype(i_c)::value;
            if constexpr (i < S_TAIL) {
                constexpr int ks = i / (2 * NF), kf = (i / NF) % 2, f = i % NF;
                if constexpr (ks == 0) aw_mfma32_first(sA[f][kf], ka[kf][0], qf[f][0], negm4[f]);
                else aw_mfma32_acc(sA[f][kf], ka[kf][1], qf[f][1]);
            } else {
                constexpr int kf = (i - S_TAIL) / NF, f = i % NF;
                aw_mfma16_acc(sA[f][kf], kt[kf], qtail[f]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");        // MFMA results -> VALU (mask / raise below)
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 6>([&](auto u_c) { k_read(u_c, std::integral_constant<int, 1>{}, k_lane + sl_cur, kt_lane + sl_cur); });
        if (32 > kv_len) mask_half(sA, 0);
        raise(sA, pbA, std::true_type{});

#ifdef AW_TIMING
        tm_loop0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- tile 0: period 0 has no PV (nothing multiplied yet)
        period(T_{}, T_{}, F_{}, F_{}, T_{}, sA, sB, pbA, pbB, k_lane + sl_next, kt_lane + sl_next, v_lane + sl_prev, 1);
        advance_ld();
        if (T > 1) {
            period(T_{}, T_{}, T_{}, T_{}, T_{}, sB, sA, pbB, pbA, k_lane + sl_next, kt_lane + sl_next, v_lane + sl_cur, 2);
            advance();
            for (int t = 1; t < T - 1; ++t) {
                period(T_{}, T_{}, T_{}, F_{}, T_{}, sA, sB, pbA, pbB, k_lane + sl_next, kt_lane + sl_next, v_lane + sl_prev, 2 * t + 1);
                advance_ld();
                if (chain && t + 5 == T) {           // the loader's next tile is tile 0 of the next unit
                    ldrsrc = unit_rsrc(nxt);
#pragma unroll
                    for (int i = 0; i < PW; ++i) doff[i] -= (unsigned)T * tile_step;
                }
                period(T_{}, T_{}, T_{}, T_{}, T_{}, sB, sA, pbB, pbA, k_lane + sl_next, kt_lane + sl_next, v_lane + sl_cur, 2 * t + 2);
                advance();
            }
            // the last tile: its odd period has no further half to score
            period(T_{}, T_{}, T_{}, F_{}, T_{}, sA, sB, pbA, pbB, k_lane + sl_next, kt_lane + sl_next, v_lane + sl_prev, 2 * T - 1);
            advance_ld();
        }
        // (the Q registers are dead from here on: the next unit's rows are requested now and arrive under the last periods)
        if (chain) q_load(nxt);
        period(F_{}, T_{}, T_{}, T_{}, T_{}, sB, sA, pbB, pbA, k_lane + sl_next, kt_lane + sl_next, v_lane + sl_cur, 2 * T);
        advance();
#ifdef AW_TIMING
        tm_loop1 = __builtin_amdgcn_s_memtime();
#endif
        // ---- the last half's PV (V half 1 of the last tile = "tile t - 1" of the period that would follow)
        period(F_{}, F_{}, T_{}, F_{}, F_{}, sA, sB, pbA, pbB, k_lane + sl_next, kt_lane + sl_next, v_lane + sl_prev, 0);
        if (chain) q_scale();           // (in front of the stores: the wait for the rows then covers loads only)
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");

        // ---- normalise and store.  A lane owns O[q = fr][d * 16 + fq * 4 .. + 3] of each fragment: stored from there a wave
        //      instruction writes 64 eight-byte pieces into 64 different lines (20 such instructions per lane: the store tail cost
        //      ~2 us per unit).  The wave's 16 NF rows go through its own 144-byte-pitch LDS area instead (above the ring, which is
        //      already receiving the next unit) and leave as whole rows, 16 bytes per lane.  Row sums: O^T[72][q] = fragment d = 4,
        //      row 8 = lane (fq = 2, fr = q), register 0.
        {
            char* const st = smem + AW_SMEM + wave * (16 * NF * AW_OPITCH);
            static_for<0, NF>([&](auto f_c) {
                constexpr int f = decltype(f_c)::value;
                f32x4 o[AW_DF];
                static_for<0, AW_DF>([&](auto d_c) { constexpr int d = decltype(d_c)::value; aw_acc_read<f * AW_DF + d>(o[d]); });
                const float inv = 1.0f / __shfl(o[AW_DF - 1][0], 32 + fr, 64);
#pragma unroll
                for (int d = 0; d < AW_DF; ++d) {
                    const int dd = d * 16 + fq * 4;
                    if (dd < AW_HD) {
                        bf16x4 ov;
#pragma unroll
                        for (int r = 0; r < 4; ++r) ov[r] = f2bf(o[d][r] * inv);
                        *reinterpret_cast<bf16x4*>(st + (f * 16 + fr) * AW_OPITCH + dd * 2) = ov;
                    }
                }
            });
            // (the wave's own LDS writes, then its own reads: in order, no barrier)
            constexpr int NCHUNK = 16 * NF * 9;                      // 16-byte chunks of the wave's rows
            bf16_t* const obase = (bf16_t*)p.out + (size_t)(cur.q_row0 + cur.qs + wave * 16 * NF) * p.ldo + cur.h * AW_HD;
            const int rows_left = cur.q_len - (cur.qs + wave * 16 * NF);
#pragma unroll
            for (int it = 0; it < (NCHUNK + 63) / 64; ++it) {
                const int c = it * 64 + lane, row = c / 9, ch = c - row * 9;
                if (c < NCHUNK && row < rows_left) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(st + row * AW_OPITCH + ch * 16);
                    if (!(AW_DBG & 64) || v[0] == 0x12345678u) *reinterpret_cast<u32x4*>(obase + (size_t)row * p.ldo + ch * 8) = v;
                }
            }
        }
#ifdef AW_TIMING
        if (lane == 0) {
            tmrec[0] = tm_start; tmrec[1] = tm_loop0; tmrec[2] = tm_loop1; tmrec[3] = __builtin_amdgcn_s_memtime();
            tmrec[4] = tm_wait; tmrec[5] = (unsigned long long)tm_nb; tmrec[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
        tm_start = __builtin_amdgcn_s_memtime(); tm_wait = 0; tm_nb = 0;
        tmrec = (unsigned long long*)p.lse + ((size_t)nxt.u * NW + wave) * 64;
#endif
        if (!nxt.ok) break;
        if (!chain) {
            // the dummy tiles this unit's loader requested past its end are still landing in the ring: drain, and nobody may be
            // reading it when the cold start overwrites it
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        cold = !chain;
        cur = nxt;
    }
}

// (AW_TIMING builds keep their stamps in the launch's `lse` buffer)
// The ViT shape only: head_dim 72, non-causal, plain q / k / v rows of one buffer (v behind k, same row stride), no extras.
bool attention72w_ok(const AttnArgs& a) {
    if (a.head_dim != AW_HD || a.causal || a.q_shared || a.kv_group > 1 || a.kv_end || a.q_in_rows || a.q_head_stride) return false;
#ifndef AW_TIMING
    if (a.lse) return false;
#endif
    if (a.ldk != a.ldv || (a.ldk & 7) || (a.ldq & 7) || (a.ldo & 3)) return false;
    // k and v must be columns of the SAME rows (the qkv GEMM's output): the kernel reads both through one descriptor over
    // [k, v + len), K-tile rows past kv_len and the 16-byte pad chunk behind a head's 144 bytes come from whatever lies in
    // between — inside one row buffer that is finite model data (a pad chunk of garbage bits would reach the tail MFMA as
    // 0 * NaN); separate allocations would be read out of bounds.  Anything else runs on attention.hip.
    const long long delta = (const char*)a.v - (const char*)a.k;
    if (delta <= 0 || delta >= (long long)a.ldk * 2) return false;
    if ((long long)a.heads * AW_HD * 2 > delta) return false;               // (the k columns end before the v columns begin)
    if (a.max_q < 192) return false;
    // 256-row query tiles: worth it when they are (nearly) full — a 1026-row image would waste a fifth of the grid
    const int full = (a.max_q + AW_QT - 1) / AW_QT * AW_QT;
    return (full - a.max_q) * 8 <= full;
}

#ifdef AW_TIMING
static unsigned long long* g_aw_timing = nullptr;
constexpr size_t AW_TIMING_UNITS = 4096;
extern "C" int vr_dbg_attn_timing_copy(void* host, size_t bytes) {
    if (!g_aw_timing) return 1;
    return (int)hipMemcpy(host, g_aw_timing, bytes < AW_TIMING_UNITS * 8 * 64 * 8 ? bytes : AW_TIMING_UNITS * 8 * 64 * 8, hipMemcpyDeviceToHost);
}
#endif

#ifndef VR_ATTN_W_WAVES
#define VR_ATTN_W_WAVES 4
#endif
// (plain kernels around the body: with launch bounds that depend on a template parameter hipcc leaves the host stub undefined)
#if VR_ATTN_W_WAVES == 8
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention72w_kernel(AttnArgs p) { attention72w_body<2, 8>(p); }
#else
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attention72w_kernel(AttnArgs p) { attention72w_body<4, 4>(p); }
#endif

hipError_t launch_attention72w(const AttnArgs& a_in, hipStream_t s) {
    AttnArgs a = a_in;
#ifdef AW_TIMING
    if (!g_aw_timing) { (void)hipMalloc((void**)&g_aw_timing, AW_TIMING_UNITS * 8 * 64 * 8); (void)hipMemset(g_aw_timing, 0, AW_TIMING_UNITS * 8 * 64 * 8); }
    a.lse = (float*)g_aw_timing;
#endif
    const int q_tiles = (a.max_q + AW_QT - 1) / AW_QT;
    constexpr int NW = VR_ATTN_W_WAVES;
    void (*k)(AttnArgs) = attention72w_kernel;
    static unsigned long long attr = 0;
    set_max_dynamic_lds((const void*)k, AW_SMEM_ALL, attr);
    // persistent: one workgroup per CU (120 KiB of LDS each) walking the units
    const int n_cu = device_cu_count();
    const int units = a.B * a.heads * q_tiles;
    hipLaunchKernelGGL(k, dim3(units < n_cu ? units : n_cu), dim3(64 * NW), AW_SMEM_ALL, s, a);
    return hipGetLastError();
}

}  // namespace vr
