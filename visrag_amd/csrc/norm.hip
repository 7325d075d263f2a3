// Wavefront LayerNorm / RMSNorm: one 64-lane wave per row, the whole row held in registers,
// fp32 statistics (two-pass mean / centred variance), bf16 output for the next MFMA GEMM.
//   LayerNorm(eps=1e-6, affine): vision_transformer.py:142,155,465,525; resampler.py:111,129-131
//   MiniCPMRMSNorm: modeling_minicpm.py:119-136 (fp32 mean-square, rsqrt(var+eps), * weight)
// Roofline: HBM (reads 4 B, writes 2 B per element).
#include "common.h"
#include "gen_math.h"
#include "kernels.h"

namespace vr {

constexpr int NORM_MAXV = 14;   // float4 per lane at most: rows up to 64*4*14 = 3584 columns (the generator's hidden size)
constexpr int NORM_STDV = 10;   // the encoder's rows (<= 2560 columns) keep the 10-register form: 14 costs its LayerNorm 14 %
constexpr int NORM_VITV = 5;    // the ViT's rows (1152 columns = 4.5 x 256): half the loop bodies of the 10-register form are masked off

// RPW rows per wave (adjacent): gamma / beta are loaded once per wave pass instead of once per row — for the ViT's 1152-column
// rows they are 9.2 KB of L1 / L2 reads per row next to the row's own 4.6 KB.
template <bool RMS, int MAXV = NORM_STDV, int RPW = 1>
__global__ __launch_bounds__(256) void norm_kernel(const float* __restrict__ x, int rows, int dim, int ldx,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ b, float eps,
                                                   bf16_t* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int nv = dim >> 2;
    f32x4 v[RPW][MAXV];
    float mu[RPW], rstd[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)min(row0 + r, rows - 1) * ldx);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + i * 64;
            v[r][i] = (c < nv) ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            if constexpr (RMS) s += v[r][i][0] * v[r][i][0] + v[r][i][1] * v[r][i][1] + v[r][i][2] * v[r][i][2] + v[r][i][3] * v[r][i][3];
            else s += v[r][i][0] + v[r][i][1] + v[r][i][2] + v[r][i][3];
        }
        s = wave_sum(s);
        mu[r] = 0.f;
        if constexpr (RMS) {
            rstd[r] = rsqrtf(s / dim + eps);
        } else {
            mu[r] = s / dim;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c = lane + i * 64;
                if (c < nv) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[r][i][e] - mu[r]; q += d * d; }
                }
            }
            q = wave_sum(q);
            rstd[r] = 1.0f / sqrtf(q / dim + eps);
        }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4 ww = reinterpret_cast<const f32x4*>(w)[c];
            f32x4 bb = {0.f, 0.f, 0.f, 0.f};
            if constexpr (!RMS) bb = reinterpret_cast<const f32x4*>(b)[c];
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                if (row0 + r < rows) {
                    f32x4 y = (v[r][i] - mu[r]) * rstd[r] * ww;
                    if constexpr (!RMS) y += bb;
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = f2bf(y[e]);
                    reinterpret_cast<bf16x4*>(out + (size_t)(row0 + r) * ldo)[c] = o;
                }
            }
        } else if (c * 4 < ldo) {
#pragma unroll
            for (int r = 0; r < RPW; ++r)
                if (row0 + r < rows) reinterpret_cast<bf16x4*>(out + (size_t)(row0 + r) * ldo)[c] = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        }
    }
}

hipError_t launch_layernorm(const float* x, int rows, int dim, int ldx, const float* w, const float* b,
                            float eps, void* out, int ldo, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (dim % 4 || ldo % 4 || ldx % 4 || ldo > 64 * 4 * NORM_MAXV || dim > ldo || dim > ldx) return hipErrorInvalidValue;
#ifndef VR_NORM_VITV
#define VR_NORM_VITV 1
#endif
    if (VR_NORM_VITV && ldo <= 64 * 4 * NORM_VITV)
        hipLaunchKernelGGL((norm_kernel<false, NORM_VITV, 2>), dim3((rows + 7) / 8), dim3(256), 0, s, x, rows, dim, ldx, w, b, eps, (bf16_t*)out, ldo);
    else if (ldo <= 64 * 4 * NORM_STDV)
        hipLaunchKernelGGL((norm_kernel<false, NORM_STDV>), dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, dim, ldx, w, b, eps, (bf16_t*)out, ldo);
    else
        hipLaunchKernelGGL((norm_kernel<false, NORM_MAXV>), dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, dim, ldx, w, b, eps, (bf16_t*)out, ldo);
    return hipGetLastError();
}

hipError_t launch_rmsnorm(const float* x, int rows, int dim, int ldx, const float* w, float eps, void* out,
                          int ldo, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (dim % 4 || ldo % 4 || ldx % 4 || ldo > 64 * 4 * NORM_MAXV || dim > ldo || dim > ldx) return hipErrorInvalidValue;
    if (ldo <= 64 * 4 * NORM_STDV)
        hipLaunchKernelGGL((norm_kernel<true, NORM_STDV>), dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, dim, ldx, w, (const float*)nullptr, eps, (bf16_t*)out, ldo);
    else
        hipLaunchKernelGGL((norm_kernel<true, NORM_MAXV>), dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, dim, ldx, w, (const float*)nullptr, eps, (bf16_t*)out, ldo);
    return hipGetLastError();
}

// Residual update fused into the RMSNorm that follows it: x <- x + alpha * (partial[0] + partial[1]
// + ...), the split-K partial products of the o / down projections (fp32, summed in a fixed
// order), then the usual RMSNorm of the new x.  One wave per row, the row stays in registers.
template <int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_accum_kernel(float* __restrict__ x, int rows, int dim, int ldx,
                                                            const float* __restrict__ partial, int nsplit,
                                                            size_t split_stride, int ldp, float alpha,
                                                            const float* __restrict__ w, float eps,
                                                            bf16_t* __restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = dim >> 2;
    f32x4* xr = reinterpret_cast<f32x4*>(x + (size_t)row * ldx);
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            f32x4 acc = reinterpret_cast<const f32x4*>(partial + (size_t)row * ldp)[c];
            for (int sp = 1; sp < nsplit; ++sp)
                acc += reinterpret_cast<const f32x4*>(partial + sp * split_stride + (size_t)row * ldp)[c];
            v[i] = xr[c] + alpha * acc;
            xr[c] = v[i];
        } else {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        s += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
    if (!out) return;
    s = wave_sum(s);
    const float rstd = rsqrtf(s / dim + eps);
    bf16_t* orow = out + (size_t)row * ldo;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + i * 64;
        if (c < nv) {
            const f32x4 y = v[i] * rstd * reinterpret_cast<const f32x4*>(w)[c];
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(y[r]);
            reinterpret_cast<bf16x4*>(orow)[c] = o;
        } else if (c * 4 < ldo) {
            reinterpret_cast<bf16x4*>(orow)[c] = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        }
    }
}

// The same for a handful of rows (the generator's decode step: ONE row, up to 64 partial planes): a workgroup per
// row, a thread per float4 column, so the planes of a column are summed by one thread and the columns run in
// parallel — the one-wave-per-row kernel above walks nsplit x dim / 256 dependent loads per lane (46 us for a
// 3584-wide row and 14 planes; this one: a few us).
__global__ __launch_bounds__(1024) void rmsnorm_accum_row_kernel(float* __restrict__ x, int dim, int ldx,
                                                                 const float* __restrict__ partial, int nsplit,
                                                                 size_t split_stride, int ldp, float alpha,
                                                                 const float* __restrict__ w, float eps,
                                                                 bf16_t* __restrict__ out, int ldo) {
    __shared__ float red[16];
    const int row = blockIdx.x, c = threadIdx.x, nv = dim >> 2;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < nv) {
        // the planes of a column are independent loads but an ordered sum: fetch eight at a time, then add in plane order
        // (a plain loop waits for every load before it asks for the next: 18 planes x ~0.3 us)
        const float* pc = partial + (size_t)row * ldp + (size_t)c * 4;
        f32x4 acc = *reinterpret_cast<const f32x4*>(pc);
        int sp = 1;
        for (; sp + 8 <= nsplit; sp += 8) {
            f32x4 t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = *reinterpret_cast<const f32x4*>(pc + (size_t)(sp + i) * split_stride);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += t[i];
        }
        for (; sp < nsplit; ++sp) acc += *reinterpret_cast<const f32x4*>(pc + (size_t)sp * split_stride);
        f32x4* xr = reinterpret_cast<f32x4*>(x + (size_t)row * ldx);
        v = xr[c] + alpha * acc;
        xr[c] = v;
    }
    if (!out) return;
    float s = wave_sum(sumsq4(v));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += red[i];          // fixed order: deterministic
    const float rstd = rsqrtf(tot / dim + eps);
    bf16_t* orow = out + (size_t)row * ldo;
    if (c < nv) {
        const f32x4 y = v * rstd * reinterpret_cast<const f32x4*>(w)[c];
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(y[r]);
        reinterpret_cast<bf16x4*>(orow)[c] = o;
    } else if (c * 4 < ldo) {
        reinterpret_cast<bf16x4*>(orow)[c] = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    }
}

hipError_t launch_rmsnorm_accum(float* x, int rows, int dim, int ldx, const float* partial, int nsplit,
                                size_t split_stride, int ldp, float alpha, const float* w, float eps, void* out,
                                int ldo, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (dim % 4 || ldx % 4 || ldp % 4 || dim > 64 * 4 * NORM_MAXV || dim > ldx || dim > ldp || nsplit < 1) return hipErrorInvalidValue;
    if (out && (ldo % 4 || ldo > 64 * 4 * NORM_MAXV || dim > ldo)) return hipErrorInvalidValue;
    if (rows <= 16 && dim <= 4096 && (!out || ldo <= 4096)) {
        hipLaunchKernelGGL(rmsnorm_accum_row_kernel, dim3(rows), dim3(1024), 0, s, x, dim, ldx, partial, nsplit, split_stride, ldp,
                           alpha, w, eps, (bf16_t*)out, ldo);
        return hipGetLastError();
    }
    if (dim <= 64 * 4 * NORM_STDV && (!out || ldo <= 64 * 4 * NORM_STDV))
        hipLaunchKernelGGL(rmsnorm_accum_kernel<NORM_STDV>, dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, dim, ldx, partial, nsplit,
                           split_stride, ldp, alpha, w, eps, (bf16_t*)out, ldo);
    else
        hipLaunchKernelGGL(rmsnorm_accum_kernel<NORM_MAXV>, dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, dim, ldx, partial, nsplit,
                           split_stride, ldp, alpha, w, eps, (bf16_t*)out, ldo);
    return hipGetLastError();
}

}  // namespace vr
