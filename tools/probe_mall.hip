// Probe: is a buffer that was just read (or just written) faster to stream again — i.e. does the memory-side cache
// (256 MB "infinity cache") serve repeat reads above HBM pace?  Streams `mb` MB with the decode GEMM's access shape
// (256 workgroups, each 16-byte loads, nt or default policy) cold (after sweeping 1 GB of other data) and warm.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_mall tools/probe_mall.hip && tools/probe_mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ __launch_bounds__(256) void stream_kernel(const char* p, size_t bytes, unsigned* sink) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7FFFFFFF, 0x00020000);
    const size_t per = bytes / gridDim.x;               // contiguous slab per workgroup
    unsigned acc = 0;
    for (size_t off = (size_t)blockIdx.x * per + threadIdx.x * 16; off < (size_t)(blockIdx.x + 1) * per; off += 256 * 16 * 8) {
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(off + (size_t)i * 4096), 0, AUX);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc ^= v[i][0] ^ v[i][3];
    }
    if (acc == 0x12345678u) *sink = acc;
}

static float run(void (*k)(const char*, size_t, unsigned*), const char* p, size_t bytes, unsigned* sink, int grid) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, p, bytes, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    char *a, *big; unsigned* sink;
    const size_t big_b = 1ull << 30;
    (void)hipMalloc(&a, 512ull << 20); (void)hipMalloc(&big, big_b); (void)hipMalloc(&sink, 4);
    (void)hipMemset(a, 1, 512ull << 20); (void)hipMemset(big, 2, big_b);
    for (int mb : {32, 64, 128, 192, 384}) {
        const size_t bytes = (size_t)mb << 20;
        for (int pol = 0; pol < 2; ++pol) {
            auto k = pol ? stream_kernel<2> : stream_kernel<0>;
            float cold = 1e9f, warm = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                run(stream_kernel<0>, big, big_b, sink, 1024);          // evict
                const float c = run(k, a, bytes, sink, 256);
                const float w = run(k, a, bytes, sink, 256);
                cold = c < cold ? c : cold; warm = w < warm ? w : warm;
            }
            printf("%4d MB, %s loads: cold %.1f us = %.2f TB/s, again %.1f us = %.2f TB/s\n", mb, pol ? "nt" : "default", cold * 1e3,
                   bytes / (cold * 1e-3) / 1e12, warm * 1e3, bytes / (warm * 1e-3) / 1e12);
        }
    }
    return 0;
}
