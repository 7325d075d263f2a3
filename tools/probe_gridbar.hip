// Probe: latency of a grid barrier over one workgroup per CU (device-scope atomics only, no fences), and whether data
// exchanged through relaxed device-scope atomic stores / loads (sc1) around it is always the fresh value across XCDs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_gridbar tools/probe_gridbar.hip && tools/probe_gridbar [wgs=256] [iters=2000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* abort_flag) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

// two-level form: 8 group counters (workgroup % 8: the XCD it was dispatched to), the last arriver of a group bumps the global one
__device__ __forceinline__ bool grid_barrier2(unsigned* counter, unsigned epoch1, unsigned* abort_flag) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned g = blockIdx.x & 7u, gsz = (gridDim.x - g + 7u) / 8u;
        const unsigned old = __hip_atomic_fetch_add(counter + 64 + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == epoch1 * gsz) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch1 * min(8u, gridDim.x);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void bar2_kernel(unsigned* counter, unsigned* abort_flag, int iters) {
    for (int it = 0; it < iters; ++it)
        if (!grid_barrier2(counter, (unsigned)(it + 1), abort_flag)) break;
}

__global__ __launch_bounds__(256) void bar_kernel(unsigned* counter, unsigned* abort_flag, float* slots, int iters, unsigned* errors,
                                                  int exchange) {
    const unsigned n = gridDim.x;
    unsigned err = 0;
    for (int it = 0; it < iters; ++it) {
        if (exchange) {
            // every thread publishes 1 float; after the barrier reads the neighbour workgroup's
            __hip_atomic_store(slots + blockIdx.x * 256 + threadIdx.x, (float)(it + 1) + blockIdx.x * 0.001f, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!grid_barrier(counter, (unsigned)(it + 1) * n, abort_flag)) break;
        if (exchange) {
            const unsigned nb = (blockIdx.x + 37) % n;
            const float v = __hip_atomic_load(slots + nb * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (float)(it + 1) + nb * 0.001f) ++err;
            // second barrier: nobody overwrites a slot before it was read
            if (!grid_barrier(counter + 32, (unsigned)(it + 1) * n, abort_flag)) break;
        }
    }
    if (err) atomicAdd(errors, err);
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned *ctr, *ab, *errs; float* slots;
    hipMalloc(&ctr, 4096); hipMalloc(&ab, 4); hipMalloc(&errs, 4); hipMalloc(&slots, wgs * 256 * 4);
    for (int exch = 0; exch < 2; ++exch) {
        hipMemset(ctr, 0, 4096); hipMemset(ab, 0, 4); hipMemset(errs, 0, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(bar_kernel, dim3(wgs), dim3(256), 0, 0, ctr, ab, slots, iters, errs, exch);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned a = 0, e = 0; hipMemcpy(&a, ab, 4, hipMemcpyDeviceToHost); hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost);
        printf("wgs %d exchange %d: %.3f us per iteration (%d barrier%s each), aborted %u, stale reads %u\n", wgs, exch,
               ms * 1e3 / iters, exch ? 2 : 1, exch ? "s" : "", a, e);
    }
    {
        hipMemset(ctr, 0, 4096); hipMemset(ab, 0, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(bar2_kernel, dim3(wgs), dim3(256), 0, 0, ctr, ab, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned a = 0; hipMemcpy(&a, ab, 4, hipMemcpyDeviceToHost);
        printf("wgs %d two-level barrier: %.3f us, aborted %u\n", wgs, ms * 1e3 / iters, a);
    }
    return 0;
}
