// Probe: what does ONE CU sustain in fp32 stores, as a function of the access shape?  (round 5: the fp32 epilogues of the
// one-wave-per-SIMD GEMM write a 256 x 192 tile in ~7.5 us = 26 GB/s per CU however many CUs are busy.)
// Every workgroup (4 waves) writes tiles of 256 rows x 768 B of a [rows][4608 B] fp32 matrix, `reps` tiles each:
//   shape 0: the epilogue's: one instruction = 16 rows x 64 B (lane = (row fr, 16-B quarter fq)), 3 column fragments x 2 halves
//   shape 1: one instruction = 4 rows x 256 B contiguous
//   shape 2: one instruction = 2 rows x 512 B contiguous ... (wraps inside the 768-B row: 1.33 rows)
//   shape 3: fully contiguous 1 KiB per instruction (a different matrix layout: tile-major)
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_store tools/probe_store.hip && tools/probe_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool LOAD>
__global__ __launch_bounds__(256) void k(float* __restrict__ out, const float* __restrict__ in, int reps, int tiles_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t LD = 1152;                         // floats per row
    f32x4 v = {1.f + lane, 2.f, 3.f, 4.f};
    for (int r = 0; r < reps; ++r) {
        const int tile = (blockIdx.x + r * gridDim.x) % tiles_total;       // 128 m-tiles x 6 n-tiles
        const int m0 = (tile / 6) * 256, n0 = (tile % 6) * 192;
        float* base = out + (size_t)(m0 + wave * 64) * LD + n0;            // this wave: 64 rows x 192 floats
        const float* ibase = in + (size_t)(m0 + wave * 64) * LD + n0;
        if (SHAPE == 0) {
            const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const size_t o = (size_t)(g * 16 + fr) * LD + j * 16 + fq * 4;
                    f32x4 x = v;
                    if (LOAD) x += *reinterpret_cast<const f32x4*>(ibase + o);
                    *reinterpret_cast<f32x4*>(base + o) = x;
                }
        } else if (SHAPE == 1) {
            const int rr = lane >> 4, c = lane & 15;
#pragma unroll
            for (int g = 0; g < 16; ++g)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const size_t o = (size_t)(g * 4 + rr) * LD + j * 64 + c * 4;
                    f32x4 x = v;
                    if (LOAD) x += *reinterpret_cast<const f32x4*>(ibase + o);
                    *reinterpret_cast<f32x4*>(base + o) = x;
                }
        } else if (SHAPE == 2) {
#pragma unroll
            for (int g = 0; g < 48; ++g) {
                const int e = g * 64 + lane;                  // 16-B element of the wave's 64 x 48 elements
                const size_t o = (size_t)(e / 48) * LD + (e % 48) * 4;
                f32x4 x = v;
                if (LOAD) x += *reinterpret_cast<const f32x4*>(ibase + o);
                *reinterpret_cast<f32x4*>(base + o) = x;
            }
        } else {
            float* b2 = out + (size_t)tile * 256 * 192 + wave * 64 * 192;
            const float* i2 = in + (size_t)tile * 256 * 192 + wave * 64 * 192;
#pragma unroll
            for (int g = 0; g < 48; ++g) {
                const size_t o = (size_t)g * 256 + lane * 4;
                f32x4 x = v;
                if (LOAD) x += *reinterpret_cast<const f32x4*>(i2 + o);
                *reinterpret_cast<f32x4*>(b2 + o) = x;
            }
        }
        v[1] += 1.f;
    }
}

template <int SHAPE, bool LOAD>
static void run(float* out, float* in, int grid, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, LOAD>), dim3(grid), dim3(256), 0, 0, out, in, reps, 768);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, LOAD>), dim3(grid), dim3(256), 0, 0, out, in, reps, 768);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * reps * 256 * 768;
    printf("shape %d %s grid %3d: %7.1f us  %6.2f us per tile  store %6.1f GB/s per CU  chip %5.2f TB/s%s\n", SHAPE, LOAD ? "rmw  " : "store",
           grid, ms * 1e3, ms * 1e3 / reps, bytes / grid / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12, LOAD ? " (+ as much read)" : "");
}

int main() {
    float *out, *in;
    const size_t n = (size_t)32768 * 1152;
    (void)hipMalloc(&out, n * 4); (void)hipMalloc(&in, n * 4);
    (void)hipMemset(out, 0, n * 4); (void)hipMemset(in, 0, n * 4);
    for (int grid : {1, 72, 256}) {
        const int reps = grid == 1 ? 64 : 24;
        run<0, false>(out, in, grid, reps); run<1, false>(out, in, grid, reps); run<2, false>(out, in, grid, reps); run<3, false>(out, in, grid, reps);
        run<0, true>(out, out, grid, reps); run<1, true>(out, out, grid, reps); run<3, true>(out, out, grid, reps);
    }
    return 0;
}
