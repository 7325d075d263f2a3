// LayerNorm folded into the GEMMs around it — the small kernels (the GEMM side is gemm256w.hip, LNF = 1 / 2).
//
//   LN(x) W^T + bias = rstd ((x o gamma) W^T) - rstd mean c1 + c2,    c1[n] = sum_k gamma[k] W[n][k],  c2 = bias + W beta
//
// (SigLIP blocks: vision_transformer.py:92-96 `x + attn(norm1(x))`, `x + mlp(norm2(x))`.)  The residual GEMM that produces
// x leaves bf16(x o gamma) (gamma applied BEFORE the one rounding: scaling the weights instead would round twice — the
// CPU emulation tests/test_cpu_ln_fold_numerics.py puts that at 1.44x the default route's error, this form at 1.08x) and, per row
// and per 96-column half tile it owns, the partial (sum, sum of squares) of the fp32 values it stored;
// ln_fold_stats_kernel combines a row's partials IN INDEX ORDER (fixed order: results do not depend on which
// workgroup finished first) in double and writes (a, b) = (rstd, -mean rstd), which the consuming GEMM's epilogue
// applies as a * acc + b * c1[n] + c2[n].  Roofline: nothing here is worth one (3 MB per launch).
// Off by default (engine.hip: VR_VIT_LN_FOLD); built in round 4 without a GPU at hand, to be measured in round 5.
#include "common.h"
#include "kernels.h"

namespace vr {

__global__ __launch_bounds__(256) void ln_fold_stats_kernel(const float* __restrict__ part, int parts, int rows, int dim, float eps,
                                                            float* __restrict__ ab) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float2* p = reinterpret_cast<const float2*>(part) + (size_t)r * parts;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < parts; ++i) { const float2 v = p[i]; s1 += (double)v.x; s2 += (double)v.y; }
    const double mean = s1 / dim;
    double var = s2 / dim - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    reinterpret_cast<float2*>(ab)[r] = float2{(float)rstd, (float)(-mean * rstd)};
}

hipError_t launch_ln_fold_stats(const float* part, int parts, int rows, int dim, float eps, float* ab, hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    if (!part || !ab || parts <= 0 || dim <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ln_fold_stats_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, part, parts, rows, dim, eps, ab);
    return hipGetLastError();
}

// one workgroup per weight row n
__global__ __launch_bounds__(256) void ln_fold_weight_kernel(const bf16_t* __restrict__ W, int k, int ldw, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ bias,
                                                             float* __restrict__ c1, float* __restrict__ c2) {
    __shared__ double red[2][256];
    const int n = blockIdx.x, tid = threadIdx.x;
    const bf16_t* w = W + (size_t)n * ldw;
    double s1 = 0.0, s2 = 0.0;
    for (int kk = tid; kk < k; kk += 256) {
        const double x = (double)bf2f(w[kk]);
        s1 += (double)gamma[kk] * x;
        s2 += (double)beta[kk] * x;
    }
    red[0][tid] = s1; red[1][tid] = s2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) { red[0][tid] += red[0][tid + st]; red[1][tid] += red[1][tid + st]; }
        __syncthreads();
    }
    if (tid == 0) {
        c1[n] = (float)red[0][0];
        c2[n] = (float)((bias ? (double)bias[n] : 0.0) + red[1][0]);
    }
}

hipError_t launch_ln_fold_weights(const void* W, int n_pad, int k, int ldw, const float* gamma, const float* beta, const float* bias,
                                  float* c1, float* c2, hipStream_t s) {
    if (n_pad <= 0) return hipSuccess;
    if (!W || !gamma || !beta || !c1 || !c2 || k <= 0 || ldw < k) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ln_fold_weight_kernel, dim3(n_pad), dim3(256), 0, s, (const bf16_t*)W, k, ldw, gamma, beta, bias, c1, c2);
    return hipGetLastError();
}

}  // namespace vr
