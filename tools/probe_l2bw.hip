// Hardware probe: how fast can a CU pull L2-resident data — (a) LDS-DMA (global_load_lds_dwordx4),
// (b) plain global_load_dwordx4 into VGPRs, (c) (b) followed by ds_write_b128 — with 8 waves per CU
// (one 512-thread workgroup per CU, like the 256^2 GEMM) and with 16.
// Working set: 16 MiB per XCD-ish slice walked repeatedly (L2/MALL resident), 128-B rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define GLBP(p) ((const __attribute__((address_space(1))) void*)(p))

template <int MODE, int LB>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, size_t span, int iters, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // each workgroup streams its own 64 KiB window per iteration (8 waves x 8 loads x 1 KiB), windows rotate over `span`
  size_t base = ((size_t)blockIdx.x * 65536) % span;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const char* p = src + base + wave * 8192 + lane * 16;
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
        __builtin_amdgcn_global_load_lds(GLBP(p + t * 1024), LDSP(smem + (((it & 1) * 65536 + wave * 8192 + t * 1024) & (LB - 1))), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      u32x4 v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = *reinterpret_cast<const u32x4*>(p + t * 1024);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (MODE == 2) *reinterpret_cast<u32x4*>(smem + (((it & 1) * 65536 + wave * 8192 + t * 1024) & (LB - 1)) + lane * 16) = v[t];
        else acc ^= v[t];
      }
    }
    base = (base + (size_t)gridDim.x * 65536) % span;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE != 1) acc[0] ^= *reinterpret_cast<uint32_t*>(smem + tid * 4);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
}

template <int MODE, int LB>
static void run(const char* name, const char* d, size_t span, int grid, uint32_t* sink) {
  const int lds = LB;
  (void)hipFuncSetAttribute((const void*)k<MODE, LB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000;
  for (int w = 0; w < 2; ++w) {
    (void)hipEventRecord(e0);
    k<MODE, LB><<<grid, 512, lds>>>(d, span, iters, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * iters * 65536.0;
  printf("%-34s grid %4d span %4zu MiB: %7.2f TB/s  (%.1f B/clk/CU at 2.0 GHz, 256 CUs)\n", name, grid, span >> 20, bytes / ms / 1e9,
         bytes / ms / 1e9 * 1e12 / 256 / 2.0e9);
}

// GEMM-like gather: a wave-load takes 8 rows x 128 B (row pitch `pitch`), a workgroup stages a 512-row x 128-B
// "K-step" (64 KiB) and then steps 128 B along the rows; PRIVATE rows per workgroup (no sharing between CUs)
// or SHARED (all workgroups of a group of `share` read the same rows, like the tiles of one GEMM row strip).
__global__ __launch_bounds__(512) void kg(const char* __restrict__ src, size_t pitch, int ksteps, int iters, int share, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const size_t rows0 = (size_t)(blockIdx.x / share) * 512;
  for (int it = 0; it < iters; ++it) {
    const int kt = it % ksteps;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int row = wave * 64 + t * 8 + (lane >> 3);
      const char* p = src + (rows0 + row) * pitch + (size_t)kt * 128 + (((lane & 7) ^ (row & 7)) << 4);
      __builtin_amdgcn_global_load_lds(GLBP(p), LDSP(smem + (it & 1) * 65536 + wave * 8192 + t * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (*reinterpret_cast<uint32_t*>(smem + tid * 4) == 0x12345u) sink[0] = 1;
}
static void rung(const char* d, size_t pitch, int share, uint32_t* sink) {
  (void)hipFuncSetAttribute((const void*)kg, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000, ksteps = (int)(pitch / 128);
  for (int w = 0; w < 2; ++w) {
    (void)hipEventRecord(e0);
    kg<<<256, 512, 131072>>>(d, pitch, ksteps, iters, share, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = 256.0 * iters * 65536.0;
  printf("gather pitch %6zu B, %3d workgroups share rows (footprint %4.0f MiB): %7.2f TB/s\n", pitch, share,
         256.0 / share * 512 * pitch / 1048576.0, bytes / ms / 1e9);
}
int main() {
  char* d; uint32_t* sink;
  const size_t cap = (size_t)2304 << 20;
  (void)hipMalloc(&d, cap + (1 << 20)); (void)hipMemset(d, 1, cap + (1 << 20)); (void)hipMalloc(&sink, 64);
  for (size_t span : {(size_t)16 << 20}) {
    run<0, 131072>("LDS-DMA dwordx4, 1 WG/CU", d, span, 256, sink);
    run<0, 65536>("LDS-DMA dwordx4, 2 WG/CU", d, span, 512, sink);
    run<1, 4096>("global_load dwordx4 -> VGPR, 1 WG/CU", d, span, 256, sink);
    run<1, 4096>("global_load -> VGPR, 2 WG/CU", d, span, 512, sink);
    run<2, 131072>("global_load -> VGPR -> ds_write, 1 WG", d, span, 256, sink);
  }
  for (size_t pitch : {(size_t)2304, (size_t)8704, (size_t)16384})
    for (int share : {1, 4, 16, 64}) rung(d, pitch, share, sink);
  return 0;
}
