// Small kernels of the EVisRAG generator's language model (gen.hip): multimodal RoPE + KV-cache append, and
// next-token selection (repetition penalty, temperature via the Gumbel-max trick, argmax).
// Arithmetic restated from the HuggingFace Qwen2.5-VL implementation (see oracle/qwen_gen_oracle.py for the
// file:line map); the reference itself delegates it to vLLM (src/evisrag/predict.py:112-123,147).
// Roofline: both are HBM/latency bound and tiny (a few KiB to ~600 KiB per call).
#include "common.h"
#include "gen_math.h"
#include "kernels.h"

namespace vr {

namespace {

// 64-bit counter hash -> uniform in (0, 1): the sampling noise of token `idx` at decode step `step`
__device__ __forceinline__ float gumbel_noise(unsigned long long seed, unsigned step, unsigned idx) {
    unsigned long long x = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)step << 32 | idx);
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    const float u = ((float)(x >> 40) + 0.5f) * (1.0f / 16777216.0f);
    return -__logf(-__logf(u));
}

}  // namespace

// One wave per (token, head slot): slots [0, H) rotate the query heads in place, [H, H + KV) rotate the key
// heads into the K cache, [H + KV, H + 2 KV) copy the value heads into the V cache.  head_dim 128: lane p holds
// the rotate-half pair (p, p + 64).  Multimodal RoPE (modeling_qwen2_5_vl.py:486-538, 557-599): the pair's angle is
// pos[c] * theta^(-2p/128) with the position component c = temporal / height / width chosen by p's section.
//   src: bf16 qkv rows [T][ld] (bias already added by the GEMM), or parts != null: fp32 split-K partial
//   planes [n_parts][plane_stride] of the same rows + bias (decode), summed here in a fixed order.
__global__ __launch_bounds__(256) void mrope_cache_kernel(const bf16_t* __restrict__ src, const float* __restrict__ parts,
                                                          int n_parts, size_t plane_stride, const float* __restrict__ bias,
                                                          int ld, int T, int H, int KV, const int* __restrict__ pos3,
                                                          int pos_stride, int sec_t, int sec_h, const float* __restrict__ inv_freq,
                                                          bf16_t* __restrict__ q_out, int ldq, bf16_t* __restrict__ k_cache,
                                                          bf16_t* __restrict__ v_cache, int ld_cache, int cache_row0,
                                                          int* __restrict__ cu_kv, const int* __restrict__ row0_dev,
                                                          const int* __restrict__ cache_rows) {
    if (row0_dev) cache_row0 = *row0_dev;              // decode steps: the cache length lives on the device (GenState::len)
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int slots = H + 2 * KV;
    if (blockIdx.x == 0 && threadIdx.x == 0 && cu_kv) { cu_kv[0] = 0; cu_kv[1] = cache_row0 + T; }
    if (slot >= T * slots) return;
    const int t = slot / slots, hs = slot % slots;
    const int col = hs * 128;
    float x1, x2;
    if (parts) {
        x1 = bias ? bias[col + lane] : 0.f;
        x2 = bias ? bias[col + 64 + lane] : 0.f;
        // independent loads, ordered sum: eight planes in flight at a time
        const float* row = parts + (size_t)t * ld + col + lane;
        int s = 0;
        for (; s + 8 <= n_parts; s += 8) {
            float a[8], b[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[i] = row[(size_t)(s + i) * plane_stride]; b[i] = row[(size_t)(s + i) * plane_stride + 64]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) { x1 += a[i]; x2 += b[i]; }
        }
        for (; s < n_parts; ++s) { x1 += row[(size_t)s * plane_stride]; x2 += row[(size_t)s * plane_stride + 64]; }
    } else {
        const bf16_t* row = src + (size_t)t * ld + col;
        x1 = bf2f(row[lane]);
        x2 = bf2f(row[64 + lane]);
    }
    if (hs < H + KV) {
        const int c = lane < sec_t ? 0 : (lane < sec_t + sec_h ? 1 : 2);
        const float pos = (float)pos3[c * pos_stride + t];
        const float ang = pos * inv_freq[lane];                                          // inv_freq[p] = theta^(-2p/128), host-made
        rope_rotate(x1, x2, cosf(ang), sinf(ang));                                       // x*cos + rotate_half(x)*sin
    }
    const size_t crow = cache_rows ? (size_t)cache_rows[t] : (size_t)(cache_row0 + t);   // (batched decode: one cache per row)
    bf16_t* dst;
    if (hs < H) dst = q_out + (size_t)t * ldq + col;
    else if (hs < H + KV) dst = k_cache + crow * ld_cache + (hs - H) * 128;
    else dst = v_cache + crow * ld_cache + (hs - H - KV) * 128;
    dst[lane] = f2bf(x1);
    dst[64 + lane] = f2bf(x2);
}

hipError_t launch_mrope_cache(const void* src_bf16, const float* parts, int n_parts, size_t plane_stride, const float* bias,
                              int ld, int T, int H, int KV, const int* pos3, int pos_stride, int sec_t, int sec_h,
                              const float* inv_freq, void* q_out, int ldq, void* k_cache, void* v_cache, int ld_cache,
                              int cache_row0, int* cu_kv, hipStream_t s, const int* row0_dev, const int* cache_rows) {
    if (T <= 0) return hipSuccess;
    const int slots = T * (H + 2 * KV);
    hipLaunchKernelGGL(mrope_cache_kernel, dim3((slots + 3) / 4), dim3(256), 0, s, (const bf16_t*)src_bf16, parts, n_parts,
                       plane_stride, bias, ld, T, H, KV, pos3, pos_stride, sec_t, sec_h, inv_freq, (bf16_t*)q_out, ldq,
                       (bf16_t*)k_cache, (bf16_t*)v_cache, ld_cache, cache_row0, cu_kv, row0_dev, cache_rows);
    return hipGetLastError();
}

// ---- the vision tower's 2-D rotary embedding (gen_vision.hip) -------------------------------------------------
// Per call: cs[t][p] = (cos, sin)(pos * freq[p]) with pos = the row's h coordinate for p < sec_h, its w coordinate
// beyond; the same for all 32 blocks and all heads, so it is computed once (R x 64 sincosf) instead of per head slot.
__global__ void rope2d_table_kernel(const int* __restrict__ pos_h, const int* __restrict__ pos_w, int T, int sec_h,
                                    const float* __restrict__ freq, float2* __restrict__ cs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * 64) return;
    const int t = i >> 6, p = i & 63;
    const float ang = (float)(p < sec_h ? pos_h[t] : pos_w[t]) * freq[p];
    cs[i] = float2{cosf(ang), sinf(ang)};
}
hipError_t launch_rope2d_table(const int* pos_h, const int* pos_w, int T, int sec_h, const float* freq, void* cs, hipStream_t s) {
    if (T <= 0) return hipSuccess;
    hipLaunchKernelGGL(rope2d_table_kernel, dim3((T * 64 + 255) / 256), dim3(256), 0, s, pos_h, pos_w, T, sec_h, freq, (float2*)cs);
    return hipGetLastError();
}

// Per block: q and k heads rotated IN PLACE in the qkv rows (head slots of 2 * hh columns, lane p < hh holds the
// rotate-half pair (p, p + hh)); v is not touched — the attention kernel reads all three straight from the qkv buffer.
// One wave per (row, head slot), slots [0, n_slots) = the q heads then the k heads (contiguous columns).
__global__ __launch_bounds__(256) void rope2d_inplace_kernel(bf16_t* __restrict__ qkv, int ld, int T, int n_slots, int hh,
                                                             const float2* __restrict__ cs) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= T * n_slots || lane >= hh) return;
    const int t = slot / n_slots, hs = slot - t * n_slots;
    bf16_t* row = qkv + (size_t)t * ld + hs * 2 * hh;
    const float2 c = cs[t * 64 + lane];
    const float x1 = bf2f(row[lane]), x2 = bf2f(row[hh + lane]);
    row[lane] = f2bf(x1 * c.x - x2 * c.y);                                               // x*cos + rotate_half(x)*sin
    row[hh + lane] = f2bf(x2 * c.x + x1 * c.y);
}
hipError_t launch_rope2d_inplace(void* qkv, int ld, int T, int n_slots, int hh, const void* cs, hipStream_t s) {
    if (T <= 0 || n_slots <= 0) return hipSuccess;
    if (hh <= 0 || hh > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rope2d_inplace_kernel, dim3((T * n_slots + 3) / 4), dim3(256), 0, s, (bf16_t*)qkv, ld, T, n_slots, hh,
                       (const float2*)cs);
    return hipGetLastError();
}

// seen[id / 32] |= 1 << (id % 32) for the prompt's token ids (the repetition penalty covers prompt and output)
__global__ void mark_seen_kernel(const int* __restrict__ ids, int n, unsigned* __restrict__ seen, int vocab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && ids[i] >= 0 && ids[i] < vocab) atomicOr(&seen[ids[i] >> 5], 1u << (ids[i] & 31));
}
hipError_t launch_mark_seen(const int* ids, int n, unsigned* seen, int vocab, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(mark_seen_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ids, n, seen, vocab);
    return hipGetLastError();
}

// Next token from one row of logits: repetition penalty on the seen ids (logit > 0 ? / penalty : * penalty — vLLM's
// and HF's RepetitionPenaltyLogitsProcessor), then argmax of logit / temperature + Gumbel noise (an exact sample of
// softmax(logit / temperature); temperature 0 = plain argmax, ties to the lowest id like torch.argmax).
// Two launches: SAMPLE_WGS workgroups reduce their slice of the vocabulary to one (score, id) key each, one wave picks
// the best of them, writes the token and marks it seen (a single workgroup over 152k logits took 94 us).
constexpr int SAMPLE_WGS = 64;

__device__ __forceinline__ unsigned long long sample_key(float l, bool seen, float penalty, float inv_t, bool noisy,
                                                         unsigned long long seed, unsigned step, unsigned i) {
    if (seen) l = l > 0.f ? l / penalty : l * penalty;
    float sc = l * inv_t;
    if (noisy) sc += gumbel_noise(seed, step, i);
    unsigned b = __float_as_uint(sc);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);                     // order-preserving map of the float
    return ((unsigned long long)b << 32) | (unsigned)(~i);              // ties: the lowest id wins
}

__global__ __launch_bounds__(256) void sample_partial_kernel(const float* __restrict__ logits, int vocab,
                                                             const unsigned* __restrict__ seen, float penalty, float temperature,
                                                             unsigned long long seed, unsigned step,
                                                             unsigned long long* __restrict__ part, const GenState* __restrict__ st) {
    __shared__ unsigned long long best[4];
    if (st) step = (unsigned)st->step;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float inv_t = temperature > 0.f ? 1.0f / temperature : 1.0f;
    unsigned long long key = 0ull;
    for (int i = blockIdx.x * 256 + tid; i < vocab; i += SAMPLE_WGS * 256) {
        const unsigned long long k = sample_key(logits[i], (seen[i >> 5] >> (i & 31)) & 1u, penalty, inv_t, temperature > 0.f, seed,
                                                step, (unsigned)i);
        key = k > key ? k : key;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(key, off, 64);
        key = o > key ? o : key;
    }
    if (lane == 0) best[wave] = key;
    __syncthreads();
    if (tid == 0) {
        unsigned long long k = best[0];
        for (int w = 1; w < 4; ++w) k = best[w] > k ? best[w] : k;
        part[blockIdx.x] = k;
    }
}

__global__ __launch_bounds__(64) void sample_final_kernel(const unsigned long long* __restrict__ part, unsigned* __restrict__ seen,
                                                          int* __restrict__ token_out, GenState* __restrict__ st, int advance) {
    unsigned long long key = part[threadIdx.x];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(key, off, 64);
        key = o > key ? o : key;
    }
    if (threadIdx.x == 0) {
        const int tok = (int)(~(unsigned)(key & 0xFFFFFFFFull));
        *token_out = tok;
        seen[tok >> 5] |= 1u << (tok & 31);
        if (st) {
            st->token = tok;                       // the next decode step may take it from here (no host round trip)
            if (advance) {                         // free-running generation: this step's row is in the cache, move on
                st->len += 1; st->step += 1;
                st->pos[0] += 1; st->pos[1] += 1; st->pos[2] += 1;
            }
        }
    }
}

// `scratch`: SAMPLE_WGS 64-bit words on the device
hipError_t launch_sample(const float* logits, int vocab, unsigned* seen, float penalty, float temperature,
                         unsigned long long seed, unsigned step, int* token_out, unsigned long long* scratch, hipStream_t s,
                         GenState* st, int step_from_state, int advance) {
    static_assert(SAMPLE_WGS == 64, "the final reduction is one wave");
    hipLaunchKernelGGL(sample_partial_kernel, dim3(SAMPLE_WGS), dim3(256), 0, s, logits, vocab, (const unsigned*)seen, penalty,
                       temperature, seed, step, scratch, (const GenState*)(step_from_state ? st : nullptr));
    hipLaunchKernelGGL(sample_final_kernel, dim3(1), dim3(64), 0, s, (const unsigned long long*)scratch, seen, token_out, st, advance);
    return hipGetLastError();
}

// Start of a decode step: the KV ranges of its attention from the cache length on the device.  One new row sees the
// whole cache (len + 1 rows); one query row x 28 heads would be 28 workgroups, so the cache is cut into up to
// GEN_ATT_SPLITS ranges (multiples of the attention kernel's 64-key tile, ~128 keys or more each) that run as
// independent "sequences" sharing the query row; ranges past the end are empty (the attention kernel skips them).
__global__ void decode_begin_kernel(GenState* __restrict__ st, int q_rows) {
    if (threadIdx.x != 0) return;
    const int L = st->len + 1;
    int splits = min(GEN_ATT_SPLITS, max(1, (L + 127) / 128));
    const int chunk = ((L + splits - 1) / splits + 63) / 64 * 64;
    splits = (L + chunk - 1) / chunk;
    st->splits = splits;
    for (int i = 0; i <= GEN_ATT_SPLITS; ++i) { st->cu_q[i] = i * q_rows; st->cu_kv[i] = min(L, i * chunk); }
}
hipError_t launch_decode_begin(GenState* st, int q_rows, hipStream_t s) {
    hipLaunchKernelGGL(decode_begin_kernel, dim3(1), dim3(64), 0, s, st, q_rows);
    return hipGetLastError();
}

// rows[row_idx[i]][:] = src[i][:]  (image embeddings dropped into the prompt's embedding rows, fp32)
__global__ void scatter_rows_kernel(const float* __restrict__ src, const int* __restrict__ row_idx, int n, int dim,
                                    float* __restrict__ dst, int ld) {
    const int i = blockIdx.x;
    for (int c = threadIdx.x; c < dim; c += blockDim.x) dst[(size_t)row_idx[i] * ld + c] = src[(size_t)i * dim + c];
}
hipError_t launch_scatter_rows(const float* src, const int* row_idx, int n, int dim, float* dst, int ld, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(n), dim3(256), 0, s, src, row_idx, n, dim, dst, ld);
    return hipGetLastError();
}

// dst[i][:] = bf16(src[row_idx[i]][:]), columns [dim, ld) zero: the vision tower's pixel rows, put into window order
// and padded to the GEMM's K on their way to bf16 (the dtype the reference's tower receives them in)
__global__ void gather_rows_bf16_kernel(const float* __restrict__ src, const int* __restrict__ row_idx, int dim,
                                        bf16_t* __restrict__ dst, int ld) {
    const int i = blockIdx.x;
    const float* row = src + (size_t)row_idx[i] * dim;
    for (int c = threadIdx.x; c < ld; c += blockDim.x) dst[(size_t)i * ld + c] = f2bf(c < dim ? row[c] : 0.f);
}
hipError_t launch_gather_rows_bf16(const float* src, const int* row_idx, int n, int dim, void* dst, int ld, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3(n), dim3(256), 0, s, src, row_idx, dim, (bf16_t*)dst, ld);
    return hipGetLastError();
}

// The image processor + the row mover in one pass (vg_vision_encode_pages): row i of dst = the bf16 patch row of
// patch (ph[i], pw[i]) of page img[i] — element (c, t, y, x) = (u8 / 255 - mean[c]) / std[c] in fp32 like the processor
// (rescale, normalise), the still image repeated over the temporal patch — columns [3 * tp * P * P, ld) zero.
struct PatchNorm { float mean[3], std[3]; };
__global__ void patch_rows_u8_kernel(const uint8_t* const* __restrict__ pages, const int* __restrict__ page_w,
                                     const int* __restrict__ img, const int* __restrict__ ph, const int* __restrict__ pw, int P,
                                     int tp, PatchNorm nm, bf16_t* __restrict__ dst, int ld) {
    const int i = blockIdx.x;
    const int im = img[i], W = page_w[im];
    const uint8_t* base = pages[im] + ((size_t)ph[i] * P * W + (size_t)pw[i] * P) * 3;
    const int pp = P * P, dim = 3 * tp * pp;
    for (int e = threadIdx.x; e < ld; e += blockDim.x) {
        float v = 0.f;
        if (e < dim) {
            const int c = e / (tp * pp), rem = e % pp, y = rem / P, x = rem % P;
            const float px = (float)base[((size_t)y * W + x) * 3 + c];
            v = (px * (1.0f / 255.0f) - nm.mean[c]) / nm.std[c];
        }
        dst[(size_t)i * ld + e] = f2bf(v);
    }
}
hipError_t launch_patch_rows_u8(const uint8_t* const* pages, const int* page_w, const int* img, const int* ph, const int* pw, int n,
                                int P, int tp, const float* mean3, const float* std3, void* dst, int ld, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    PatchNorm nm;
    for (int c = 0; c < 3; ++c) { nm.mean[c] = mean3[c]; nm.std[c] = std3[c]; }
    hipLaunchKernelGGL(patch_rows_u8_kernel, dim3(n), dim3(256), 0, s, pages, page_w, img, ph, pw, P, tp, nm, (bf16_t*)dst, ld);
    return hipGetLastError();
}

// one decode step's attention was run as S independent KV ranges (attention_kernel with q_shared), the `group` query
// heads of a KV head as the rows of a tile (layout: kernels.h, SkinnyCombine): merge the ranges.  (The decode step itself
// merges inside its o projection — gemm_skinny.hip, COMBINE; this kernel serves K ranges longer than that kernel's ring.)
__global__ void attn_combine_kernel(const bf16_t* __restrict__ part, const float* __restrict__ lse, int S, int heads, int group,
                                    bf16_t* __restrict__ out, const int* __restrict__ S_dev, int ld_out) {
    const int row = blockIdx.y;                       // batched decode: sequence `row` of the step
    if (S_dev) S = S_dev[row];
    const int h = blockIdx.x, d = threadIdx.x;        // 128 threads: one per channel of the head
    const int kvh = heads / group, hkv = h / group, g = h % group;
    part += (size_t)row * GEN_ATT_SPLITS * heads * 128;
    lse += (size_t)row * GEN_ATT_SPLITS * heads;
    out += (size_t)row * ld_out;
    // every load up front (S <= GEN_ATT_SPLITS): the sums below then run without waiting on memory
    float l[GEN_ATT_SPLITS], pv[GEN_ATT_SPLITS];
#pragma unroll
    for (int s = 0; s < GEN_ATT_SPLITS; ++s) {
        const int r = (min(s, max(S - 1, 0)) * group + g) * kvh + hkv;
        l[s] = s < S ? lse[r] : -INFINITY;
        pv[s] = bf2f(part[(size_t)r * 128 + d]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < GEN_ATT_SPLITS; ++s) mx = fmaxf(mx, l[s]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int s = 0; s < GEN_ATT_SPLITS; ++s) {
        const float w = exp2f(l[s] - mx);
        merge_range(num, den, w, pv[s]);
    }
    out[h * 128 + d] = f2bf(num / den);
}
hipError_t launch_attn_combine(const void* part, const float* lse, int S, int heads, int group, void* out, hipStream_t s,
                               const int* S_dev, int n_rows, int ld_out) {
    if (group <= 0 || heads % group || n_rows <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(attn_combine_kernel, dim3(heads, n_rows), dim3(128), 0, s, (const bf16_t*)part, lse, S, heads, group,
                       (bf16_t*)out, S_dev, ld_out);
    return hipGetLastError();
}

}  // namespace vr
