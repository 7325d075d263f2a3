// Synthetic page images on the device — bench / test support, no reference counterpart.
//
// BASELINE config 3 is "100k synthetic page corpus: embed + HBM-resident index + 1k-query top-10": 100 000 DISTINCT
// pages have to come from somewhere, and visrag_amd/synth.py::synth_pages (numpy, ~20 ms a page) would take half an
// hour of host time.  This kernel produces the SAME pages, bit for bit (tests/test_gpu_pipeline.py), from the same
// integer hash: a coarse colour mosaic, dark text-like bars whose layout depends on the page id, +-7 pixel noise.
// One thread per pixel; the bars of a page (<= 40) are derived once per workgroup into LDS.
#include "common.h"
#include "kernels.h"

namespace vr {

__device__ __forceinline__ uint64_t synth_mix64(uint64_t x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t synth_h(uint64_t pid, uint64_t i, uint64_t salt) {
    return synth_mix64(i * 0x9E3779B97F4A7C15ull + pid * 0xD1B54A32D192ED03ull + salt);
}

__global__ __launch_bounds__(256) void synth_pages_kernel(uint8_t* __restrict__ out, int n, int size, long long seed,
                                                          long long first) {
    __shared__ int bar[40][5];          // y0, y1, x0, x1, dark
    __shared__ int nbars_s, cell_s;
    const int page = blockIdx.y;
    const uint64_t pid = (uint64_t)((first + page) * 1000003ll + seed * 7919ll + 17ll);
    if (threadIdx.x == 0) {
        cell_s = 16 + (int)(synth_h(pid, 0, 1) % 5ull) * 16;
        nbars_s = 5 + (int)(synth_h(pid, 0, 2) % 36ull);
    }
    if (threadIdx.x < 40) {
        const uint64_t b = threadIdx.x;
        const int y0 = (int)(synth_h(pid, b, 3) % (uint64_t)(size - 8));
        const int hh = 3 + (int)(synth_h(pid, b, 4) % 7ull);
        const int x0 = (int)(synth_h(pid, b, 5) % (uint64_t)(size - 40));
        const int ww = 20 + (int)(synth_h(pid, b, 6) % (uint64_t)(size - x0 - 20));
        bar[b][0] = y0; bar[b][1] = y0 + hh; bar[b][2] = x0; bar[b][3] = x0 + ww;
        bar[b][4] = 1 + (int)(synth_h(pid, b, 7) % 7ull);
    }
    __syncthreads();
    const int cell = cell_s, nbars = nbars_s;
    const int64_t npix = (int64_t)size * size;
    for (int64_t px = (int64_t)blockIdx.x * 256 + threadIdx.x; px < npix; px += (int64_t)gridDim.x * 256) {
        const int y = (int)(px / size), x = (int)(px % size);
        const uint64_t cid = (uint64_t)((y / cell) * 64 + (x / cell));
        int v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = 150 + (int)(synth_h(pid, cid, 10 + c) % 100ull);
        for (int b = 0; b < nbars; ++b) {
            if (y >= bar[b][0] && y < bar[b][1] && x >= bar[b][2] && x < bar[b][3]) {
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c] = v[c] * bar[b][4] / 16;
            }
        }
        const int noise = (int)(synth_h(pid, (uint64_t)px, 8) % 15ull) - 7;
        uint8_t* o = out + ((size_t)page * npix + px) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (uint8_t)min(255, max(0, v[c] + noise));
    }
}

hipError_t launch_synth_pages(uint8_t* out, int n, int size, long long seed, long long first, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (size < 64 || size > 4096) return hipErrorInvalidValue;
    const int bx = (int)std::min<int64_t>(((int64_t)size * size + 255) / 256, 1024);
    hipLaunchKernelGGL(synth_pages_kernel, dim3(bx, n), dim3(256), 0, s, out, n, size, seed, first);
    return hipGetLastError();
}

}  // namespace vr
