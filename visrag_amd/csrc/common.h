// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of libvisrag_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define VR_WAVE 64

#define VR_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define VR_GLOBAL(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware block remap (8 XCDs, block b is dispatched to XCD b % 8): give every XCD a
// contiguous range of logical tiles so neighbouring tiles share operand panels in that
// XCD's private L2.  Bijective for any nwg (cdna_hip_programming.md section 5, "XCD swizzle
// must be bijective").
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// Kernels with more than 64 KiB of dynamic LDS: raise the function's limit once per DEVICE (the attribute lives in
// the device's copy of the module; one process may drive several devices).  `done` is the call site's static mask.
inline void set_max_dynamic_lds(const void* fn, int bytes, unsigned long long& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
    if (!((done >> dev) & 1ull)) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        done |= 1ull << dev;
    }
}
// CU count of the current device (cached per device id)
inline int device_cu_count() {
    static int cu_of[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cu_of[dev]) {
        hipDeviceProp_t prop;
        cu_of[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cu_of[dev];
}
// counted wait + raw barrier: lets LDS-DMA loads younger than the N-th stay in flight across the
// barrier (__syncthreads() would drain the whole queue)
#define VR_WAIT_VM_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier" ::: "memory")
