// Weight-preparation kernels (run once at load time): dtype conversion + zero padding +
// row re-mapping (q|k|v stacking, 16-row gate/up interleave for the SwiGLU epilogue, transpose
// of resampler.proj which the reference applies as `x @ proj`, resampler.py:167).
#include "common.h"
#include "pack.h"

namespace vr {

template <typename SrcT>
__global__ __launch_bounds__(256) void pack_weight_kernel(const SrcT* __restrict__ src, int rows, int cols,
                                                          int src_ld, int transpose, bf16_t* __restrict__ dst,
                                                          int dst_ld, int blk, int blk_stride, int blk_off, int lo_part) {
    // dst[map(r)][c] = transpose ? src[c][r] : src[r][c]   (lo_part: what bf16 rounding left over, v - bf16(v))
    const size_t n = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        float v = transpose ? (float)src[(size_t)c * src_ld + r] : (float)src[(size_t)r * src_ld + c];
        if (lo_part) v -= bf2f(f2bf(v));
        const int dr = (r / blk) * blk_stride + blk_off + (r % blk);
        dst[(size_t)dr * dst_ld + c] = f2bf(v);
    }
}

hipError_t launch_pack_weight(const void* src, int src_is_bf16, int rows, int cols, int src_ld,
                              int transpose, void* dst, int dst_ld, int blk, int blk_stride, int blk_off,
                              hipStream_t s, int lo_part) {
    const size_t n = (size_t)rows * cols;
    if (n == 0) return hipSuccess;
    const int blocks = (int)min((size_t)4096, (n + 255) / 256);
    if (src_is_bf16)
        hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)src, rows,
                           cols, src_ld, transpose, (bf16_t*)dst, dst_ld, blk, blk_stride, blk_off, lo_part);
    else
        hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)src, rows,
                           cols, src_ld, transpose, (bf16_t*)dst, dst_ld, blk, blk_stride, blk_off, lo_part);
    return hipGetLastError();
}

// The same with a block map on the columns too: dst[rmap(r)][cmap(c)] = src[r][c], map(x) = (x / blk) * stride + x % blk.
// (The EVisRAG vision tower gives every head_dim-80 head a 128-wide slot: 40-row blocks of q / k / v at a stride of 64
// rows, and the matching 40-column blocks of the output projection.)
template <typename SrcT>
__global__ __launch_bounds__(256) void pack_weight_blocks_kernel(const SrcT* __restrict__ src, int rows, int cols, int src_ld,
                                                                 bf16_t* __restrict__ dst, int dst_ld, int rblk, int rstride,
                                                                 int cblk, int cstride) {
    const size_t n = (size_t)rows * cols;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i % cols);
        const int dr = (r / rblk) * rstride + (r % rblk), dc = (c / cblk) * cstride + (c % cblk);
        dst[(size_t)dr * dst_ld + dc] = f2bf((float)src[(size_t)r * src_ld + c]);
    }
}

hipError_t launch_pack_weight_blocks(const void* src, int src_is_bf16, int rows, int cols, int src_ld, void* dst, int dst_ld,
                                     int rblk, int rstride, int cblk, int cstride, hipStream_t s) {
    const size_t n = (size_t)rows * cols;
    if (n == 0) return hipSuccess;
    if (rblk <= 0 || cblk <= 0) return hipErrorInvalidValue;
    const int blocks = (int)min((size_t)4096, (n + 255) / 256);
    if (src_is_bf16)
        hipLaunchKernelGGL(pack_weight_blocks_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)src, rows, cols, src_ld,
                           (bf16_t*)dst, dst_ld, rblk, rstride, cblk, cstride);
    else
        hipLaunchKernelGGL(pack_weight_blocks_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)src, rows, cols, src_ld,
                           (bf16_t*)dst, dst_ld, rblk, rstride, cblk, cstride);
    return hipGetLastError();
}

template <typename SrcT>
__global__ __launch_bounds__(256) void pack_patch_weight_kernel(const SrcT* __restrict__ src, int D, int P,
                                                                bf16_t* __restrict__ dst, int dst_ld) {
    const int K = 3 * P * P;
    const size_t n = (size_t)D * K;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / K), k = (int)(i % K);          // k = c*P*P + ky*P + kx (conv weight order)
        const int c = k / (P * P), ky = (k % (P * P)) / P, kx = k % P;
        dst[(size_t)r * dst_ld + ky * 3 * P + kx * 3 + c] = f2bf((float)src[i]);
    }
}

hipError_t launch_pack_patch_weight(const void* src, int src_is_bf16, int D, int P, void* dst, int dst_ld, hipStream_t s) {
    const size_t n = (size_t)D * 3 * P * P;
    if (n == 0) return hipSuccess;
    const int blocks = (int)min((size_t)4096, (n + 255) / 256);
    if (src_is_bf16) hipLaunchKernelGGL(pack_patch_weight_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)src, D, P, (bf16_t*)dst, dst_ld);
    else hipLaunchKernelGGL(pack_patch_weight_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)src, D, P, (bf16_t*)dst, dst_ld);
    return hipGetLastError();
}

template <typename SrcT>
__global__ void to_f32_kernel(const SrcT* __restrict__ src, float* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dst[i] = (float)src[i];
}

// dst[(i / blk) * stride + off + i % blk] = src[i]  (bias vectors that follow a block-mapped weight)
template <typename SrcT>
__global__ void to_f32_blocks_kernel(const SrcT* __restrict__ src, float* __restrict__ dst, size_t n, int blk, int stride, int off) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dst[(i / blk) * stride + off + i % blk] = (float)src[i];
}

hipError_t launch_to_f32_blocks(const void* src, int src_is_bf16, float* dst, size_t n, int blk, int stride, int off, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (blk <= 0) return hipErrorInvalidValue;
    const int blocks = (int)min((size_t)4096, (n + 255) / 256);
    if (src_is_bf16) hipLaunchKernelGGL(to_f32_blocks_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)src, dst, n, blk, stride, off);
    else hipLaunchKernelGGL(to_f32_blocks_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)src, dst, n, blk, stride, off);
    return hipGetLastError();
}

hipError_t launch_to_f32(const void* src, int src_is_bf16, float* dst, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const int blocks = (int)min((size_t)4096, (n + 255) / 256);
    if (src_is_bf16) hipLaunchKernelGGL(to_f32_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)src, dst, n);
    else hipLaunchKernelGGL(to_f32_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)src, dst, n);
    return hipGetLastError();
}

}  // namespace vr
