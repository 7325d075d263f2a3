// HBM-bound glue kernels of the encode path: token-embedding
// gather, final RMSNorm + weighted-mean pool + L2 normalise, dtype conversions.
#include "common.h"
#include "kernels.h"

namespace vr {

// ---- K13: embed_tokens(ids) * scale_emb (modeling_minicpmv.py:139-141) ------------------------
__global__ __launch_bounds__(256) void embed_gather_kernel(const int* __restrict__ ids, int T,
                                                           const bf16_t* __restrict__ table, int dim,
                                                           float scale, float* __restrict__ out) {
    const int t = blockIdx.x;
    const bf16_t* src = table + (size_t)ids[t] * dim;
    float* dst = out + (size_t)t * dim;
    for (int c = threadIdx.x * 4; c < dim; c += 1024) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(src + c);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = bf2f(v[r]) * scale;
        *reinterpret_cast<f32x4*>(dst + c) = o;
    }
}

hipError_t launch_embed_gather(const int* ids, int T, const void* table, int dim, float scale,
                               float* out, hipStream_t s) {
    if (T <= 0) return hipSuccess;
    if (dim % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(embed_gather_kernel, dim3(T), dim3(256), 0, s, ids, T, (const bf16_t*)table, dim,
                       scale, out);
    return hipGetLastError();
}

// ---- K19+K20: final RMSNorm, wmean pool (w_t = t+1), L2 normalise ----------------------------
// modeling_minicpm.py:1280 (self.norm), dense_retrieval_model.py:180-184 (wmean in fp32),
// :222-223 (F.normalize, eps 1e-12).  One workgroup per sequence; wave w takes tokens
// w, w+4, ...; a lane keeps dim/64 running sums in registers.
constexpr int POOL_MAXV = 10;

// mode: 0 wmean (w_t = t + 1), 1 mean, 2 lasttoken (last_token_pool, right padding: dense_retrieval_model.py:26-34),
// 3 cls (hidden[:, 0]: :217-218) — the deterministic poolings of DRModel.encode (:172-220)
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ h,
                                                   const int* __restrict__ seq_offsets, int dim,
                                                   const float* __restrict__ norm_w, float eps,
                                                   float* __restrict__ out, float* __restrict__ tap, int mode) {
    __shared__ float red[4][64 * 4 * POOL_MAXV];
    __shared__ float red_s[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t0 = seq_offsets[b], L = seq_offsets[b + 1] - t0;
    const int nv = dim >> 2;
    f32x4 acc[POOL_MAXV], ww[POOL_MAXV];
#pragma unroll
    for (int i = 0; i < POOL_MAXV; ++i) {
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = lane + i * 64;
        ww[i] = (c < nv) ? reinterpret_cast<const f32x4*>(norm_w)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int t = wave; t < L; t += 4) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(h + (size_t)(t0 + t) * dim);
        f32x4 v[POOL_MAXV];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < POOL_MAXV; ++i) {
            const int c = lane + i * 64;
            v[i] = (c < nv) ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
            ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
        }
        ss = wave_sum(ss);
        const float rstd = rsqrtf(ss / dim + eps);
        const float wt = mode == 0 ? (float)(t + 1) : mode == 1 ? 1.0f : mode == 2 ? (t == L - 1 ? 1.0f : 0.0f) : (t == 0 ? 1.0f : 0.0f);
#pragma unroll
        for (int i = 0; i < POOL_MAXV; ++i) {
            const f32x4 y = v[i] * rstd * ww[i];
            acc[i] += y * wt;
            const int c = lane + i * 64;
            if (tap && c < nv) reinterpret_cast<f32x4*>(tap + (size_t)(t0 + t) * dim)[c] = y;
        }
    }
#pragma unroll
    for (int i = 0; i < POOL_MAXV; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(lane + i * 64) * 4 + r] = acc[i][r];
    __syncthreads();
    const float denom = mode == 0 ? 0.5f * (float)L * (float)(L + 1) : mode == 1 ? (float)L : 1.0f;   // sum of the weights
    float sq = 0.f;
    float vals[POOL_MAXV];
#pragma unroll
    for (int i = 0; i < POOL_MAXV; ++i) {
        const int c = threadIdx.x + i * 256;
        float v = 0.f;
        if (c < dim) v = (red[0][c] + red[1][c] + red[2][c] + red[3][c]) / denom;
        vals[i] = v;
        sq += v * v;
    }
    sq = wave_sum(sq);
    if (lane == 0) red_s[wave] = sq;
    __syncthreads();
    const float nrm = fmaxf(sqrtf(red_s[0] + red_s[1] + red_s[2] + red_s[3]), 1e-12f);
#pragma unroll
    for (int i = 0; i < POOL_MAXV; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < dim) out[(size_t)b * dim + c] = vals[i] / nrm;
    }
}

hipError_t launch_pool(const float* h, const int* seq_offsets, int B, int dim, const float* norm_w,
                       float eps, float* out, float* tap_hidden, hipStream_t s, int mode) {
    if (B <= 0) return hipSuccess;
    if (dim % 4 || dim > 64 * 4 * POOL_MAXV || mode < 0 || mode > 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pool_kernel, dim3(B), dim3(256), 0, s, h, seq_offsets, dim, norm_w, eps, out,
                       tap_hidden, mode);
    return hipGetLastError();
}

// ---- conversions -------------------------------------------------------------------------------
__global__ void f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, size_t n) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + i);
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
        *reinterpret_cast<bf16x4*>(out + i) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t j = n & ~(size_t)3; j < n; ++j) out[j] = f2bf(in[j]);
}

hipError_t launch_f32_to_bf16(const float* in, void* out, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const int blocks = (int)min((size_t)2048, (n / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks), dim3(256), 0, s, in, (bf16_t*)out, n);
    return hipGetLastError();
}

// f32 -> bf16 of the first n elements, zeros up to n_total (query rows + their padding in one launch;
// n and n_total multiples of 4)
__global__ void f32_to_bf16_pad_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, size_t n, size_t n_total,
                                       int* __restrict__ zero_word) {
    if (zero_word && blockIdx.x == 0 && threadIdx.x < 2) zero_word[threadIdx.x] = 0;   // (the search's two flag counters: saves a memset launch)
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (; i < n_total; i += stride) {
        bf16x4 o = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        if (i < n) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(in + i);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
        }
        *reinterpret_cast<bf16x4*>(out + i) = o;
    }
}

hipError_t launch_f32_to_bf16_pad(const float* in, void* out, size_t n, size_t n_total, hipStream_t s, int* zero_word) {
    if (n_total == 0) return hipSuccess;
    if ((n | n_total) & 3) return hipErrorInvalidValue;
    const int blocks = (int)min((size_t)2048, (n_total / 4 + 255) / 256);
    hipLaunchKernelGGL(f32_to_bf16_pad_kernel, dim3(blocks), dim3(256), 0, s, in, (bf16_t*)out, n, n_total, zero_word);
    return hipGetLastError();
}

__global__ void split_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ hi,
                                  bf16_t* __restrict__ lo, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float v = in[i];
        const bf16_t h = f2bf(v);
        hi[i] = h;
        lo[i] = f2bf(v - bf2f(h));
    }
}

hipError_t launch_split_bf16(const float* in, void* hi, void* lo, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const int blocks = (int)min((size_t)2048, (n + 255) / 256);
    hipLaunchKernelGGL(split_bf16_kernel, dim3(blocks), dim3(256), 0, s, in, (bf16_t*)hi, (bf16_t*)lo, n);
    return hipGetLastError();
}

// position of every packed token inside its own sequence (RoPE position ids = arange(L),
// modeling_minicpm.py:1203-1211)
__global__ void iota_pos_kernel(const int* __restrict__ seq_offsets, int* __restrict__ pos) {
    const int b = blockIdx.x;
    const int t0 = seq_offsets[b], L = seq_offsets[b + 1] - t0;
    for (int t = threadIdx.x; t < L; t += blockDim.x) pos[t0 + t] = t;
}

hipError_t launch_iota_pos(const int* seq_offsets, int B, int* pos, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(iota_pos_kernel, dim3(B), dim3(256), 0, s, seq_offsets, pos);
    return hipGetLastError();
}

// *flag |= 1 if any of the n 16-bit words is non-zero (load time: is the low half of a weight split empty?)
__global__ void any_nonzero16_kernel(const uint16_t* __restrict__ p, size_t n, int* __restrict__ flag) {
    int any = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        any |= (p[i] & 0x7FFFu) != 0;
    if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

hipError_t launch_any_nonzero16(const void* p, size_t n, int* flag, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const int blocks = (int)min((size_t)2048, (n + 255) / 256);
    hipLaunchKernelGGL(any_nonzero16_kernel, dim3(blocks), dim3(256), 0, s, (const uint16_t*)p, n, flag);
    return hipGetLastError();
}

}  // namespace vr
