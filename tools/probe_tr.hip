// Hardware probe (not part of the library): semantics of ds_read_b64_tr_b16 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* addrs, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((char*)lds + addrs[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x*4+j] = r[j];
}
int main() {
  int *d_a; short* d_o; hipMalloc(&d_a, 256); hipMalloc(&d_o, 512);
  for (int pat = 0; pat < 6; ++pat) {
    std::vector<int> a(64);
    for (int l = 0; l < 64; ++l) {
      int i = l & 15, g = l >> 4;
      switch (pat) {
        case 0: a[l] = 0; break;
        case 1: a[l] = l * 8; break;
        case 2: a[l] = i * 8 + g * 1024; break;
        case 3: a[l] = (i / 4) * 256 + (i % 4) * 8 + g * 2048; break;     // lane i -> row i/4, colquad i%4
        case 4: a[l] = (i % 4) * 256 + (i / 4) * 8 + g * 2048; break;     // lane i -> row i%4, colquad i/4
        case 5: a[l] = i * 512 + g * 8; break;                            // every lane its own row
      }
    }
    hipMemcpy(d_a, a.data(), 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d_a, d_o);
    std::vector<short> o(256);
    hipMemcpy(o.data(), d_o, 512, hipMemcpyDeviceToHost);
    printf("pattern %d (addr in elements: lane0..15 =", pat);
    for (int l = 0; l < 16; ++l) printf(" %d", a[l] / 2);
    printf(")\n");
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d: %5d %5d %5d %5d%s", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3], (l % 4 == 3) ? "\n" : " |");
    }
  }
  return 0;
}
