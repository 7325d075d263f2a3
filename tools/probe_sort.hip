// Hardware probe: the DPP / permlane-swap lane exchanges and the 64-lane sort of wave_sort.h
// against __shfl_xor and a host sort; and the cost of one sort in both forms.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "../visrag_amd/csrc/wave_sort.h"
using namespace vr;

__device__ uint64_t shx(uint64_t v, int m) {
  return ((uint64_t)__shfl_xor((uint32_t)(v >> 32), m, 64) << 32) | (uint32_t)__shfl_xor((uint32_t)v, m, 64);
}
__device__ uint64_t old_sort(uint64_t key, int lane) {
  for (int k = 2; k <= 64; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      const uint64_t other = shx(key, j);
      const bool keep_max = (((lane & k) == 0) == ((lane & j) == 0));
      const uint64_t mx = key > other ? key : other, mn = key > other ? other : key;
      key = keep_max ? mx : mn;
    }
  return key;
}
template <bool NEW>
__global__ void sort_k(const uint64_t* in, uint64_t* out, int reps) {
  const int lane = threadIdx.x & 63;
  uint64_t v = in[blockIdx.x * blockDim.x + threadIdx.x];
  for (int r = 0; r < reps; ++r) {
    v = NEW ? wave_sort_desc(v) : old_sort(v, lane);
    if (r + 1 < reps) v = v * 6364136223846793005ull + 1442695040888963407ull;   // rescramble
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main() {
  const int n = 256 * 256;
  std::vector<uint64_t> h(n);
  uint64_t s = 88172645463325252ull;
  for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = s; }
  for (int i = 0; i < 64; i += 3) h[64 + i] = h[64];           // duplicates in the second wave
  uint64_t *din, *dout; int* dbad;
  (void)hipMalloc(&din, n * 8); (void)hipMalloc(&dout, n * 8); (void)hipMalloc(&dbad, 64 * 4);
  (void)hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice);
  std::vector<uint64_t> o(n);
  for (int nw = 0; nw < 2; ++nw) {
    if (nw) sort_k<true><<<256, 256>>>(din, dout, 1); else sort_k<false><<<256, 256>>>(din, dout, 1);
    (void)hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost);
    int wrong = 0;
    for (int w = 0; w < n / 64; ++w) {
      std::vector<uint64_t> ref(h.begin() + w * 64, h.begin() + w * 64 + 64);
      std::sort(ref.begin(), ref.end(), std::greater<uint64_t>());
      for (int i = 0; i < 64; ++i) wrong += ref[i] != o[w * 64 + i];
    }
    printf("%s sort: %d wrong keys\n", nw ? "dpp" : "bpermute", wrong);
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int nw = 0; nw < 2; ++nw) {
    for (int it = 0; it < 2; ++it) {
      (void)hipEventRecord(e0);
      if (nw) sort_k<true><<<256, 256>>>(din, dout, 1000); else sort_k<false><<<256, 256>>>(din, dout, 1000);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.1f ns per 64-key sort per wave (1 wave per SIMD)\n", nw ? "dpp" : "bpermute", ms * 1e6 / 1000);
  }
  return 0;
}
