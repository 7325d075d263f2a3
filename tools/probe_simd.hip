// Hardware probe: which SIMD does wave i of a 512-thread workgroup land on? (HW_REG_HW_ID)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out) {
  extern __shared__ char smem[];
  unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, offset 0, size 32
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 64 * 8 * 4);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  k<<<64, 512, 128 * 1024>>>(d);
  unsigned h[64 * 8]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 12; ++b) {
    printf("block %2d:", b);
    for (int w = 0; w < 8; ++w) printf("  w%d simd=%u wave=%u cu=%u", w, (h[b*8+w] >> 4) & 3, h[b*8+w] & 15, (h[b*8+w] >> 8) & 15);
    printf("\n");
  }
  return 0;
}
