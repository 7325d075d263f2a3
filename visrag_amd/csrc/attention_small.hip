// Self-attention of SHORT packed sequences — the MiniCPM decoder over a page's 68 tokens (36 heads x 64, causal;
// modeling_minicpm.py:895-903), 40 launches per encode step.
//
// attention.hip walks 64-key tiles with a workgroup of four waves per (sequence, head) and 128 query rows per
// workgroup: for 68 tokens that is two key tiles (the second holds 4 keys), a third of the waves without a row,
// K / V staged through LDS behind two barriers per tile — 16 us per launch (18.7 in the model) for 40 MB of traffic
// and 0.35 GFLOP.  Here a workgroup of FIVE waves owns a (sequence, head) of up to 80 tokens and wave w the 16-query
// block w (all 5 760 waves of the decoder's launch are resident at once):
//   * K and Q fragments come straight from global memory in MFMA operand order (16 bytes per lane: row = fragment
//     row, 8 consecutive d); a wave asks for its Q block and the key blocks the causal mask leaves it, all up front;
//   * S^T = K Q^T for those key blocks (the transposed form of attention.hip: a lane owns a query column, max / sum are
//     in-lane + two permlane swaps), softmax over ALL keys at once (no online rescaling: 80 keys fit the registers), P
//     packed to bf16 as the B operand of the second MFMA;
//   * V goes through ONE LDS tile per workgroup (row-major, 160-byte pitch, rows past the sequence zero; the only
//     barrier of the kernel) to be read back TRANSPOSED (ds_read_b64_tr_b16) as the A operand of O^T += V^T P^T — the
//     key permutation of that read is the one the packed P carries (attention_body.h).
// 66 MFMAs per (sequence, head) instead of 176.  (First version: ONE wave per (sequence, head) walking its five query
// blocks — correct, but 18 us per launch: a serial chain of ~8 us per wave with one or two waves on a SIMD.)
// Roofline: HBM (q | k | v rows in, one output row out).
#include "attention_body.h"

namespace vr {

namespace {
constexpr int AS_BLOCKS = 5;                 // 16-row blocks: sequences of up to 80 tokens
constexpr int AS_ROWS = 96;                  // V rows in LDS: three 32-key steps of the transposing read
constexpr int AS_PITCH = 160;                // bytes per V row (128 of data): conflict-free for the tr reads (attention.hip)
constexpr int AS_WAVES = AS_BLOCKS;          // one wave per 16-query block
}  // namespace

bool attention_small_ok(const AttnArgs& a) {
    return a.head_dim == 64 && a.max_q <= 16 * AS_BLOCKS && a.cu_q == a.cu_kv && !a.q_shared && a.kv_group <= 1 && !a.kv_end &&
           !a.q_in_rows && !a.q_head_stride && !a.lse;
}

__global__ __launch_bounds__(64 * AS_WAVES) void attention_small_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char Vt[AS_ROWS * AS_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int unit = blockIdx.x;
    const int h = unit % p.heads, b = unit / p.heads;
    const int row0 = p.cu_q[b], L = p.cu_q[b + 1] - row0;
    if (L <= 0) return;                                      // (workgroup-uniform)
    const int nb = (L + 15) >> 4;                            // blocks in use
    const float sc = p.scale * 1.44269504088896340736f;      // exp2 domain
    const bf16_t* qbase = (const bf16_t*)p.q + (size_t)row0 * p.ldq + h * 64;
    const bf16_t* kbase = (const bf16_t*)p.k + (size_t)row0 * p.ldk + h * 64;
    const bf16_t* vbase = (const bf16_t*)p.v + (size_t)row0 * p.ldv + h * 64;
    const int qb = wave;                                     // this wave's query block
    const int q = qb * 16 + fr;
    const int kmax_w = p.causal ? qb : nb - 1;               // last key block this wave needs

    // ---- requested first: V rows (for the LDS tile: ten 16-byte chunks per 160-byte row, eight of data, two of padding),
    //      this wave's Q fragments and the K fragments of its key blocks — lane holds X[row = blk*16 + fr][d = ks*32 + fq*8 .. +7]
    u32x4 vst[AS_ROWS * 10 / (64 * AS_WAVES)];
#pragma unroll
    for (int it = 0; it < AS_ROWS * 10 / (64 * AS_WAVES); ++it) {
        const int c = it * (64 * AS_WAVES) + tid, row = c / 10, ch = c % 10;
        vst[it] = u32x4{0u, 0u, 0u, 0u};
        if (row < L && ch < 8) vst[it] = *reinterpret_cast<const u32x4*>(vbase + (size_t)row * p.ldv + ch * 8);
    }
    bf16x8 qf[2], ka[AS_BLOCKS][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        u32x4 raw = {0u, 0u, 0u, 0u};
        if (q < L) raw = *reinterpret_cast<const u32x4*>(qbase + (size_t)q * p.ldq + ks * 32 + fq * 8);
        qf[ks] = __builtin_bit_cast(bf16x8, raw);
    }
#pragma unroll
    for (int kb = 0; kb < AS_BLOCKS; ++kb) {
        const int key = kb * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 raw = {0u, 0u, 0u, 0u};
            if (kb <= kmax_w && key < L) raw = *reinterpret_cast<const u32x4*>(kbase + (size_t)key * p.ldk + ks * 32 + fq * 8);
            ka[kb][ks] = __builtin_bit_cast(bf16x8, raw);
        }
    }
#pragma unroll
    for (int it = 0; it < AS_ROWS * 10 / (64 * AS_WAVES); ++it) {
        const int c = it * (64 * AS_WAVES) + tid, row = c / 10, ch = c % 10;
        *reinterpret_cast<u32x4*>(Vt + row * AS_PITCH + ch * 16) = vst[it];
    }
    __syncthreads();                                         // the V tile is complete (the kernel's only barrier)
    if (qb >= nb) return;
    const int v_off = (fq * 4 + (fr >> 2)) * AS_PITCH + (fr & 3) * 8;
    {
        const int kmax = kmax_w;
        // ---- S^T[key][q] for the visible key blocks; the others stay at -inf (P = 0)
        f32x4 s[AS_BLOCKS + 1];
#pragma unroll
        for (int kb = 0; kb <= AS_BLOCKS; ++kb) s[kb] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kb = 0; kb < AS_BLOCKS; ++kb) {
            if (kb > kmax) continue;                         // wave-uniform
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[kb][0], qf[0], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[kb][1], qf[1], a, 0, 0, 0);
            const int lim = p.causal ? min(L - 1, q) : L - 1;                  // last visible key of this lane's query
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kb][r] = (kb * 16 + fq * 4 + r > lim) ? -INFINITY : a[r];
        }
        // ---- softmax over all keys of the query column (in-lane + the four lanes of the column)
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < AS_BLOCKS; ++kb) mx = fmaxf(mx, fmaxf(fmaxf(s[kb][0], s[kb][1]), fmaxf(s[kb][2], s[kb][3])));
        mx = col4_max(mx);
        const float neg_m = (mx == -INFINITY) ? 0.f : -mx * sc;               // (rows past the sequence: nothing visible)
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < AS_BLOCKS; ++kb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kb][r] = __builtin_amdgcn_exp2f(fmaf(s[kb][r], sc, neg_m));
            rs += (s[kb][0] + s[kb][1]) + (s[kb][2] + s[kb][3]);
        }
        s[AS_BLOCKS] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float l = col4_sum(rs);
        // ---- O^T[d][q] += V^T P^T over 32-key steps: P^T = (block 2 kp | block 2 kp + 1) packed, V^T by the transposing read
        f32x4 o[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < (AS_BLOCKS + 1) / 2; ++kp) {
            if (2 * kp > kmax) continue;                     // wave-uniform
            bf16x8 pb;
#pragma unroll
            for (int r = 0; r < 4; ++r) { pb[r] = f2bf(s[2 * kp][r]); pb[4 + r] = f2bf(s[2 * kp + 1][r]); }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const char* vr = Vt + v_off + kp * 32 * AS_PITCH + d * 32;
                const bf16x8 va = __builtin_shufflevector(lds_tr_read(vr), lds_tr_read(vr + 16 * AS_PITCH), 0, 1, 2, 3, 4, 5, 6, 7);
                o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, o[d], 0, 0, 0);
            }
        }
        // ---- normalise and store: lane owns out[q][h*64 + d*16 + fq*4 .. +3]
        if (q < L) {
            const float inv = 1.0f / l;
            bf16_t* orow = (bf16_t*)p.out + (size_t)(row0 + q) * p.ldo + h * 64;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                bf16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = f2bf(o[d][r] * inv);
                *reinterpret_cast<bf16x4*>(orow + d * 16 + fq * 4) = ov;
            }
        }
    }
}

hipError_t launch_attention_small(const AttnArgs& a, hipStream_t s) {
    const int units = a.B * a.heads;
    if (units <= 0) return hipSuccess;
    hipLaunchKernelGGL(attention_small_kernel, dim3(units), dim3(64 * AS_WAVES), 0, s, a);
    return hipGetLastError();
}

}  // namespace vr
