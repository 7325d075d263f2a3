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
// a 2-step cross-lane reduction (v_permlane16_swap / v_permlane32_swap: register to register, no
// LDS crossbar), and the final O row is 4 consecutive d per lane (8-byte stores).
// head_dim 72: QK^T = 3 x (16x16x32) MFMA per fragment (d padded with zeros), PV = 5 d-fragments
// (80; column 72 of V holds 1.0, so the PV MFMA also produces the softmax row sums).  Row pitch
// 160 B (288 B for head_dim 128) = 8 mod 64 dwords makes both the b128 K reads and the tr_b16 V
// reads bank-conflict free.
//
// Pipeline (template PIPE):
//   0  one K/V slot, two barriers per tile (kept for head_dim 128: two slots would not fit the
//      static LDS limit; the resampler call is < 0.2 % of an encode step)
//   1  two K/V slots: tile t+1 is written into the other slot at the top of iteration t and the
//      global loads of tile t+2 are issued right behind it — ONE barrier per tile, no LDS
//      write -> barrier -> read chain on the critical path
//   2  as 1, and the score MFMAs run one tile AHEAD: S(t+1) = K[t+1] Q^T is issued before the
//      softmax of tile t, so the exp/convert VALU work of a wave overlaps its own matrix work
//      instead of relying on the co-resident wave (K slots therefore lead the V slots by a tile)
//   3  as 1, but the tiles are staged by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
//      instruction, five per wave and tile) instead of registers + ds_write_b128: no staging
//      VGPRs (-24: three workgroups per CU instead of two), no LDS-write instructions (the
//      160-byte pitch makes ds_write_b128 2-way conflicted: 37 % extra LDS cycles, probe_lds.hip)
//      and no vmcnt stall in front of the writes.  The DMA image is lane-linear; lanes that would
//      hit a row's padding chunk are masked off, so the zeros / 1.0 column written once stay put.
// PIPE 0..2: K / V tiles come in through buffer loads whose descriptor ends at the sequence's last
// row: rows past kv_len read as zeros (no address clamping, no per-tile 64-bit address arithmetic).
// Roofline: MFMA (4*N^2*D flop per head).
#include "attention_body.h"

namespace vr {

template <int HD, int QF, int PIPE>
__global__ __launch_bounds__(256, QF >= 4 ? 1 : (PIPE == 3 ? 3 : 2)) void attention_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[attn_smem_bytes<HD, PIPE>()];
    const int q_tiles = (p.max_q + 64 * QF - 1) / (64 * QF);
    attention_body<HD, QF, PIPE>(p, xcd_remap(blockIdx.x, p.B * p.heads * q_tiles), smem);
}

#ifndef VR_ATTN_PIPE
#define VR_ATTN_PIPE 3
#endif
#ifndef VR_ATTN_QF
#define VR_ATTN_QF 2
#endif

template <int HD, int QF, int PIPE>
static hipError_t launch_t(const AttnArgs& a, hipStream_t s) {
    const int q_tiles = (a.max_q + 64 * QF - 1) / (64 * QF);
    hipLaunchKernelGGL((attention_kernel<HD, QF, PIPE>), dim3(a.B * a.heads * q_tiles), dim3(256), 0, s, a);
    return hipGetLastError();
}

#ifndef VR_ATTN_SMALL
#define VR_ATTN_SMALL 1
#endif
bool attention_small_ok(const AttnArgs& a);
hipError_t launch_attention_small(const AttnArgs& a, hipStream_t s);
// the ViT shape on the one-wave-per-SIMD kernel with the hand-ordered stream (attention_w.hip)
#ifndef VR_ATTN_W
#define VR_ATTN_W 1
#endif
bool attention72w_ok(const AttnArgs& a);
hipError_t launch_attention72w(const AttnArgs& a, hipStream_t s);

hipError_t launch_attention(const AttnArgs& a_in, hipStream_t s) {
    if (a_in.B <= 0 || a_in.max_q <= 0) return hipSuccess;
    if ((a_in.ldq | a_in.ldk | a_in.ldv) % 8 || a_in.ldo % 4) return hipErrorInvalidValue;
    AttnArgs a = a_in;
    // q rows that already carry scale * log2(e): attention_w.hip takes them as they are, the other kernels multiply the scores by
    // scale * log2(e) themselves — hand them the scale that makes that factor 1
    if (a.q_prescaled && !(VR_ATTN_W && attention72w_ok(a))) a.scale = 1.0f / 1.44269504088896340736f;
    // self-attention of short packed sequences (the decoder over a page's 68 tokens): a wave per (sequence, head)
    if (VR_ATTN_SMALL && attention_small_ok(a)) return launch_attention_small(a, s);
    if (VR_ATTN_W && attention72w_ok(a)) return launch_attention72w(a, s);
    // q-fragments per wave: 2 (32 rows) for long sequences; PIPE 0 (single slot, no pipeline prologue) is the
    // fastest form for the one- or two-tile sequences of the decoder (68-token pages: 13.8 vs 15.7 us)
    const bool big = a.max_q > 64;
    const bool tiny = a.max_q <= 128;
    switch (a.head_dim) {
        case 64:
            if (tiny) return big ? launch_t<64, 2, 0>(a, s) : launch_t<64, 1, 0>(a, s);
            return launch_t<64, VR_ATTN_QF, VR_ATTN_PIPE>(a, s);
        case 72:
            if (tiny) return big ? launch_t<72, 2, 0>(a, s) : launch_t<72, 1, 0>(a, s);
            return launch_t<72, VR_ATTN_QF, VR_ATTN_PIPE>(a, s);
        case 80:
            if (tiny) return big ? launch_t<80, 2, 0>(a, s) : launch_t<80, 1, 0>(a, s);
            return launch_t<80, VR_ATTN_QF, VR_ATTN_PIPE>(a, s);
        case 128: return launch_t<128, 1, 0>(a, s);
        default:  return hipErrorInvalidValue;
    }
}

}  // namespace vr
