// Probe: the chip clocks to its power budget (GEMM K-loops run at ~1.73 GHz here, not 2.4) — so which MFMA shape does more
// work per joule?  One wave per SIMD, 4 waves per CU, every CU: a long loop of independent accumulate chains on RANDOM bf16
// operands held in registers (no memory traffic), v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_32x32x16_bf16, plus a variant
// that also re-reads its operands from LDS at the GEMM's rate (16 ds_read_b128 per 64 MFMAs of the 16x16x32 form).
// Reports sustained TFLOP/s over ~60 ms (long enough for the clock to settle) and the shader clock from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_mfma_power tools/probe_mfma_power.hip && tools/probe_mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool LDS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k(const u32x4* __restrict__ src, float* sink, int iters,
                                                                                    unsigned long long* clk) {
    __shared__ u32x4 lds[4096];                       // 64 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
    __syncthreads();
    bf16x8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, src[(i * 64 + lane) & 4095]);
        b[i] = __builtin_bit_cast(bf16x8, src[(2048 + i * 64 + lane) & 4095]);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (SHAPE == 0) {
        f32x4 acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            if (LDS) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    a[i] = __builtin_bit_cast(bf16x8, lds[(it * 16 + i * 64 + lane + wave * 512) & 4095]);
                    b[i] = __builtin_bit_cast(bf16x8, lds[(it * 16 + 2048 + i * 64 + lane + wave * 512) & 4095]);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
        if (s == 12345.678f) sink[0] = s;
    } else {
        // the same 128 x 128 x 32 of work per iteration: 4 x 4 tiles of 32 x 32, two k-halves of 16
        f32x16 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            if (LDS) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    a[i] = __builtin_bit_cast(bf16x8, lds[(it * 16 + i * 64 + lane + wave * 512) & 4095]);
                    b[i] = __builtin_bit_cast(bf16x8, lds[(it * 16 + 2048 + i * 64 + lane + wave * 512) & 4095]);
                }
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i * 2 + kk], b[j * 2 + kk], acc[i][j], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
        if (s == 12345.678f) sink[0] = s;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int SHAPE, bool LDS>
static void run(const u32x4* src, float* sink, unsigned long long* clk, const char* name) {
    const int iters = 60000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(256), dim3(256), 0, 0, src, sink, iters, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double flop = 2.0 * 128 * 128 * 32 * 8 / 8.0;      // per iteration and wave: 64 MFMAs of 16x16x32 (= 32 of 32x32x16)
        const double total = (double)iters * 256 * 4 * 2.0 * 128 * 128 * 32 * 2;   // two k-halves of 32... (see below)
        (void)flop; (void)total;
        const double fl = (double)iters * 256.0 * 4.0 * 64.0 * (2.0 * 16 * 16 * 32);
        printf("%-28s run %d: %7.2f ms  %7.1f TFLOP/s  shader clock %.0f MHz\n", name, rep, ms, fl / (ms * 1e-3) / 1e12,
               (double)h[0] / ((double)h[1] / 100.0));
    }
}

int main() {
    u32x4* src; float* sink; unsigned long long* clk;
    (void)hipMalloc(&src, 4096 * 16); (void)hipMalloc(&sink, 64); (void)hipMalloc(&clk, 64);
    uint16_t* h = (uint16_t*)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 8; ++i) {           // random bf16 in (-2, 2): random sign, exponent 125..127, random mantissa
        const unsigned r = (unsigned)rand();
        h[i] = (uint16_t)(((r & 1) << 15) | ((125 + (r >> 1) % 3) << 7) | ((r >> 8) & 0x7F));
    }
    (void)hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
    run<0, false>(src, sink, clk, "16x16x32 registers");
    run<1, false>(src, sink, clk, "32x32x16 registers");
    run<0, true>(src, sink, clk, "16x16x32 + 16 LDS reads");
    run<1, true>(src, sink, clk, "32x32x16 + 16 LDS reads");
    (void)hipMemset(src, 0, 4096 * 16);
    run<0, false>(src, sink, clk, "16x16x32 zeros");
    run<1, false>(src, sink, clk, "32x32x16 zeros");
    return 0;
}
