// Probe: how many VALU fillers does a lone wave hide behind one MFMA, by MFMA shape and by where its A / B operands live
// (arch VGPRs or accumulation registers)?     hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_fill.hip -o tools/probe_mfma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define FILL1 "v_fma_f32 %[f0], %[f0], %[c], %[c]\n\t"
#define FILL2 FILL1 "v_fma_f32 %[f1], %[f1], %[c], %[c]\n\t"
#define FILL3 FILL2 "v_fma_f32 %[f2], %[f2], %[c], %[c]\n\t"
#define FILL4 FILL3 "v_fma_f32 %[f3], %[f3], %[c], %[c]\n\t"
#define FILL5 FILL4 "v_fma_f32 %[f4], %[f4], %[c], %[c]\n\t"
#define FILL6 FILL5 "v_fma_f32 %[f5], %[f5], %[c], %[c]\n\t"
#define EXP1 "v_exp_f32 %[f0], %[f0]\n\t"
#define EXP2 EXP1 "v_exp_f32 %[f1], %[f1]\n\t"
#define EXP3 EXP2 "v_exp_f32 %[f2], %[f2]\n\t"

// MODE 0: 16x16x32, A/B VGPR; 1: 16x16x32, A/B AGPR; 2: 32x32x16 A/B VGPR; 3: 32x32x16 A/B AGPR.  D/C always AGPR (8 accumulators).
template <int MODE, int NF, int KIND>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(unsigned long long* out, int iters) {
    float f0 = threadIdx.x, f1 = 1.f, f2 = 2.f, f3 = 3.f, f4 = 4.f, f5 = 5.f, c = 0.5f;
    u32x4 a = {1, 2, 3, 4}, b = {5, 6, 7, 8};
    asm volatile("v_accvgpr_write_b32 a200, %0\n\tv_accvgpr_write_b32 a201, %0\n\tv_accvgpr_write_b32 a202, %0\n\tv_accvgpr_write_b32 a203, %0\n\t"
                 "v_accvgpr_write_b32 a204, %0\n\tv_accvgpr_write_b32 a205, %0\n\tv_accvgpr_write_b32 a206, %0\n\tv_accvgpr_write_b32 a207, %0"
                 : : "v"(c) : "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define STEP(ACC16, ACC32) \
        if constexpr (MODE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 " ACC16 ", %[a], %[b], " ACC16 "\n\t" : : [a] "v"(a), [b] "v"(b)); \
        if constexpr (MODE == 1) asm volatile("v_mfma_f32_16x16x32_bf16 " ACC16 ", a[200:203], a[204:207], " ACC16 "\n\t" : :); \
        if constexpr (MODE == 2) asm volatile("v_mfma_f32_32x32x16_bf16 " ACC32 ", %[a], %[b], " ACC32 "\n\t" : : [a] "v"(a), [b] "v"(b)); \
        if constexpr (MODE == 3) asm volatile("v_mfma_f32_32x32x16_bf16 " ACC32 ", a[200:203], a[204:207], " ACC32 "\n\t" : :); \
        if constexpr (KIND == 0) { \
            if constexpr (NF == 1) asm volatile(FILL1 : [f0] "+v"(f0) : [c] "v"(c)); \
            if constexpr (NF == 2) asm volatile(FILL2 : [f0] "+v"(f0), [f1] "+v"(f1) : [c] "v"(c)); \
            if constexpr (NF == 3) asm volatile(FILL3 : [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2) : [c] "v"(c)); \
            if constexpr (NF == 4) asm volatile(FILL4 : [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3) : [c] "v"(c)); \
            if constexpr (NF == 5) asm volatile(FILL5 : [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [f4] "+v"(f4) : [c] "v"(c)); \
            if constexpr (NF == 6) asm volatile(FILL6 : [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [f4] "+v"(f4), [f5] "+v"(f5) : [c] "v"(c)); \
        } else { \
            if constexpr (NF == 1) asm volatile(EXP1 : [f0] "+v"(f0)); \
            if constexpr (NF == 2) asm volatile(EXP2 : [f0] "+v"(f0), [f1] "+v"(f1)); \
            if constexpr (NF == 3) asm volatile(EXP3 : [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2)); \
        }
        STEP("a[0:3]", "a[0:15]") STEP("a[4:7]", "a[16:31]") STEP("a[8:11]", "a[32:47]") STEP("a[12:15]", "a[48:63]")
        STEP("a[16:19]", "a[64:79]") STEP("a[20:23]", "a[80:95]") STEP("a[24:27]", "a[96:111]") STEP("a[28:31]", "a[112:127]")
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("" : : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5) : "a0", "a127");
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (f0 + f1 + f2 + f3 + f4 + f5 == 12345.678f) out[1] = 1;
}

template <int MODE, int NF, int KIND>
static void run(unsigned long long* d, const char* name) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<MODE, NF, KIND>), dim3(256), dim3(256), 0, 0, d, iters);
    hipLaunchKernelGGL((probe<MODE, NF, KIND>), dim3(256), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-34s %d %s fillers/MFMA: %6.1f ticks per MFMA\n", name, NF, KIND ? "v_exp" : "v_fma", (double)h / (iters * 8.0));
}

int main() {
    unsigned long long* d; hipMalloc(&d, 64);
#define ROW(MODE, NAME) run<MODE, 0, 0>(d, NAME); run<MODE, 1, 0>(d, NAME); run<MODE, 2, 0>(d, NAME); run<MODE, 3, 0>(d, NAME); run<MODE, 4, 0>(d, NAME); \
                        run<MODE, 5, 0>(d, NAME); run<MODE, 6, 0>(d, NAME); run<MODE, 1, 1>(d, NAME); run<MODE, 2, 1>(d, NAME); run<MODE, 3, 1>(d, NAME);
    ROW(0, "16x16x32  A/B in VGPRs") ROW(1, "16x16x32  A/B in AGPRs") ROW(2, "32x32x16  A/B in VGPRs") ROW(3, "32x32x16  A/B in AGPRs")
    return 0;
}
