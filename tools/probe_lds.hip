// Hardware probe (not part of the library): LDS bank-conflict cycles of the attention kernel's access
// patterns, one pattern per kernel, for several row pitches.  Run under
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace -- tools/probe_lds
// and read the counters per kernel name (probe<PATTERN, PITCH>).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// PAT 0: K fragment reads (ds_read_b128): row kf*16+fr, byte ks*64 + fq*16
// PAT 1: K tail reads (ds_read_b64): row kf*16+fr, byte 128 + fq*8
// PAT 2: V transposing reads (ds_read_b64_tr_b16): row (fq*4 + fr/4) (+16), byte (fr%4)*8 + d*32
// PAT 3: staging writes (ds_write_b128): chunk c = tid + i*256 -> row c/9, byte (c%9)*16
// PAT 4: V transposing reads with the (fr%4, fr/4) roles swapped: row fq*4 + fr%4, byte (fr/4)*8 + d*32
template <int PAT, int PITCH>
__global__ __launch_bounds__(256) void probe(unsigned* out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 64 * PITCH + 4096];
    const int tid = threadIdx.x, lane = tid & 63;
    const int fr = lane & 15, fq = lane >> 4;
    for (int i = tid; i < (int)sizeof(smem) / 4; i += 256) ((unsigned*)smem)[i] = i;
    __syncthreads();
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        if constexpr (PAT == 0) {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const u32x4 v = *(const volatile u32x4*)(smem + (kf * 16 + fr) * PITCH + ks * 64 + fq * 16);
                    acc += v[0] ^ v[3];
                }
        } else if constexpr (PAT == 1) {
#pragma unroll
            for (int kf = 0; kf < 4; ++kf) {
                const u32x2 v = *(const volatile u32x2*)(smem + (kf * 16 + fr) * PITCH + 128 + fq * 8);
                acc += v[0] ^ v[1];
            }
        } else if constexpr (PAT == 2 || PAT == 4) {
            const int off = PAT == 2 ? (fq * 4 + (fr >> 2)) * PITCH + (fr & 3) * 8 : (fq * 4 + (fr & 3)) * PITCH + (fr >> 2) * 8;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int d = 0; d < 5; ++d)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        s16x4 r;
                        asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r)
                                     : "v"((unsigned)(size_t)(smem + off + ks * 32 * PITCH + h * 16 * PITCH + d * 32) ) : "memory");
                        acc += (unsigned)r[0] ^ (unsigned)r[3];
                    }
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                int c = tid + i * 256; if (c > 64 * 9 - 1) c = 64 * 9 - 1;
                *(volatile u32x4*)(smem + (c / 9) * PITCH + (c % 9) * 16) = u32x4{acc, 1, 2, 3};
                *(volatile u32x4*)(smem + 64 * PITCH + (c / 9) * PITCH + (c % 9) * 16) = u32x4{acc, 1, 2, 3};
            }
        }
    }
    out[blockIdx.x * 256 + tid] = acc;
}

template <int PAT, int PITCH>
static void run(unsigned* d) {
    hipLaunchKernelGGL((probe<PAT, PITCH>), dim3(512), dim3(256), 0, 0, d, 200);
    hipDeviceSynchronize();
}

int main() {
    unsigned* d;
    hipMalloc(&d, 512 * 256 * 4);
    run<0, 160>(d); run<1, 160>(d); run<2, 160>(d); run<3, 160>(d); run<4, 160>(d);
    run<0, 144>(d); run<2, 144>(d); run<3, 144>(d);
    run<0, 176>(d); run<2, 176>(d); run<3, 176>(d);
    run<0, 208>(d); run<2, 208>(d);
    run<0, 272>(d); run<2, 272>(d);
    printf("done\n");
    return 0;
}
