// Bicubic image resize on the GPU, BIT-EXACT with Pillow's Image.resize(size, BICUBIC) on 8-bit RGB
// — the resize the MiniCPM-V slicing policy applies to every non-448x448 page
// (modeling_minicpmv.py:497-527: `image.resize(best_size, Image.Resampling.BICUBIC)`).
//
// Pillow's algorithm (src/libImaging/Resample.c, third-party; restated from its published source):
//   * separable two-pass, HORIZONTAL first into an 8-bit intermediate, then VERTICAL;
//   * per output coordinate a window [xmin, xmin+n) and n double-precision weights
//       center = (xx + 0.5) * scale, scale = in/out, support = 2.0 * max(scale, 1),
//       w(x) = bicubic_{a=-0.5}((x + xmin - center + 0.5) / max(scale,1)), normalised to sum 1;
//   * weights converted to 22-bit fixed point, round-half-away-from-zero;
//   * pixel = clip8((2^21 + sum_x in[x] * k[x]) >> 22)   (arithmetic shift, clamp to 0..255).
// The coefficient tables are tiny and computed on the host in double precision exactly like
// Pillow does; the two passes are HBM-bound byte kernels (one thread per output pixel, 3 channels).
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "kernels.h"

namespace vr {

constexpr int RS_PRECISION_BITS = 32 - 8 - 2;

__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ in, int in_w, int rows,
                                                       uint8_t* __restrict__ out, int out_w,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk,
                                                       int ksize) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * out_w) return;
    const int y = idx / out_w, xx = idx % out_w;
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int* k = kk + (size_t)xx * ksize;
    const uint8_t* src = in + ((size_t)y * in_w + xmin) * 3;
    int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
        const int w = k[x];
        s0 += src[3 * x + 0] * w;
        s1 += src[3 * x + 1] * w;
        s2 += src[3 * x + 2] * w;
    }
    uint8_t* dst = out + (size_t)idx * 3;
    dst[0] = (uint8_t)min(max(s0 >> RS_PRECISION_BITS, 0), 255);
    dst[1] = (uint8_t)min(max(s1 >> RS_PRECISION_BITS, 0), 255);
    dst[2] = (uint8_t)min(max(s2 >> RS_PRECISION_BITS, 0), 255);
}

__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ in, int w, uint8_t* __restrict__ out,
                                                       int out_h, const int* __restrict__ bounds,
                                                       const int* __restrict__ kk, int ksize) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= out_h * w) return;
    const int yy = idx / w, x = idx % w;
    const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int* k = kk + (size_t)yy * ksize;
    int s0 = 1 << (RS_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < n; ++y) {
        const uint8_t* src = in + ((size_t)(ymin + y) * w + x) * 3;
        const int wgt = k[y];
        s0 += src[0] * wgt;
        s1 += src[1] * wgt;
        s2 += src[2] * wgt;
    }
    uint8_t* dst = out + (size_t)idx * 3;
    dst[0] = (uint8_t)min(max(s0 >> RS_PRECISION_BITS, 0), 255);
    dst[1] = (uint8_t)min(max(s1 >> RS_PRECISION_BITS, 0), 255);
    dst[2] = (uint8_t)min(max(s2 >> RS_PRECISION_BITS, 0), 255);
}

static inline double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// Pillow precompute_coeffs + normalize_coeffs_8bpc
int resize_coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& kk) {
    const double scale = (double)in_size / (double)out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    bounds.assign((size_t)out_size * 2, 0);
    kk.assign((size_t)out_size * ksize, 0);
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (int x = 0; x < xmax; ++x) {
            const double v = k[x] * (double)(1 << RS_PRECISION_BITS);
            kk[(size_t)xx * ksize + x] = (v < 0) ? (int)(-0.5 + v) : (int)(0.5 + v);
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}

hipError_t launch_resize_h(const uint8_t* in, int in_w, int rows, uint8_t* out, int out_w, const int* bounds,
                           const int* kk, int ksize, hipStream_t s) {
    const int n = rows * out_w;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(resize_h_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, in_w, rows, out, out_w, bounds, kk, ksize);
    return hipGetLastError();
}

hipError_t launch_resize_v(const uint8_t* in, int w, uint8_t* out, int out_h, const int* bounds, const int* kk,
                           int ksize, hipStream_t s) {
    const int n = out_h * w;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(resize_v_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, w, out, out_h, bounds, kk, ksize);
    return hipGetLastError();
}

}  // namespace vr
