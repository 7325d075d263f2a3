// 256 x 256 x 64 tile GEMM on v_mfma_f32_32x32x16_bf16.
//
// Same tile, staging, pipeline and fused epilogues as gemm256_bf16_kernel (gemm.hip), but the wave's
// 128 (M) x 64 (N) sub-tile is computed with 32x32x16 MFMAs: 2 (N) x 4 (M) accumulators of 16 fp32,
// 8 MFMAs per 16-deep k-step fed by 6 ds_read_b128.  The 32x32 shape has the higher matrix-pipe
// ceiling on gfx950 (2.38-2.49 PF vs 2.07 PF for 16x16x32, MI355X_MICROARCH.md) and issues half
// as many matrix instructions per flop.
//
// Operands stay swapped (W = MFMA "A", activations = "B"):  D[n][m], lane l holds
//     m = m_base + (l & 31),   n = n_base + 8*(reg>>2) + 4*(l>>5) + (reg&3)      reg = 0..15
// i.e. again groups of FOUR CONSECUTIVE OUTPUT COLUMNS per lane (vector epilogue), 8 groups per
// 32-row m-fragment: column offset of group (jn, g) = jn*32 + g*8 + (l>>5)*4.
// LDS image: 128-byte rows, 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7): conflict-free for
// the 32-row fragment reads (the (r & 7) swizzle of the 16x16 kernels is 2-way here).
#include "gemm_core.h"
#include "gemm_epilogue.h"
#include "lab.h"
#include "gemm_core_lab.h"

namespace vr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void stage_glds32(const bf16_t* __restrict__ g, int ld, int row0, int k0, char* tile,
                                             int wave, int lane) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int rbase = wave * 32 + t * 8;
        const int row = rbase + (lane >> 3);
        const int kc = (lane & 7) ^ ((row >> 1) & 7);
        const bf16_t* src = g + (size_t)(row0 + row) * ld + k0 + kc * 8;
        __builtin_amdgcn_global_load_lds(VR_GLOBAL(src), VR_LDS(tile + rbase * 128), 16, 0, 0);
    }
}

// Epilogue for one lane-row m of the 32x32 layout: v[jn*4 + g] = 4 consecutive columns at
// nb + jn*32 + g*8 + h*4 (h = lane >> 5).  Mirrors gemm_epilogue_row (16x16 layout).
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_row32(f32x4 (&v)[8], const GemmArgs& p, int m, int nb, int h) {
    if (m >= p.M) return;
    const int orow = p.rowmap ? p.rowmap[m] : m;
    if (orow < 0) return;
    if constexpr (EPI == EPI_SWIGLU) {
        // packed W rows: [16 gate | 16 up] blocks -> within a 32-column fragment groups g = 0,1 are
        // gate and g = 2,3 the matching up columns
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            if (nb + jn * 32 >= p.N) continue;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int oc = (nb + jn * 32) / 2 + g * 8 + h * 4;
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(silu(v[jn * 4 + g][r]) * v[jn * 4 + g + 2][r]);
                *reinterpret_cast<bf16x4*>((bf16_t*)p.out + (size_t)orow * p.ldo + oc) = o;
            }
        }
    } else if constexpr (EPI == EPI_ROPE) {
        if (nb >= p.N) return;
        if (nb < p.rope_cols) {   // head = the wave's 64 columns; pair (c, c+32) = (jn 0, jn 1), same g
            const float* tab = p.rope_table + (size_t)p.rope_pos[m] * 64;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 cs = *reinterpret_cast<const f32x4*>(tab + g * 8 + h * 4);
                const f32x4 sn = *reinterpret_cast<const f32x4*>(tab + 32 + g * 8 + h * 4);
                const f32x4 x1 = v[g], x2 = v[4 + g];
                v[g] = x1 * cs - x2 * sn;
                v[4 + g] = x2 * cs + x1 * sn;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = nb + (i >> 2) * 32 + (i & 3) * 8 + h * 4;
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(v[i][r]);
            *reinterpret_cast<bf16x4*>((bf16_t*)p.out + (size_t)orow * p.ldo + n) = o;
        }
    } else {
        const float* rb = nullptr;
        if (p.rowbias) rb = p.rowbias + (size_t)(m % p.rowbias_period) * p.rowbias_ld;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = nb + (i >> 2) * 32 + (i & 3) * 8 + h * 4;
            if (n >= p.N) continue;
            f32x4 x = v[i];
            if (p.bias) x += *reinterpret_cast<const f32x4*>(p.bias + n);
            if (rb && n < p.rowbias_cols) x += *reinterpret_cast<const f32x4*>(rb + n);
            if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(EPI == EPI_GELU ? gelu_erf(x[r]) : x[r]);
                *reinterpret_cast<bf16x4*>((bf16_t*)p.out + (size_t)orow * p.ldo + n) = o;
            } else if constexpr (EPI == EPI_F32) {
                *reinterpret_cast<f32x4*>((float*)p.out + (size_t)orow * p.ldo + n) = x;
            } else {
                const f32x4 rs = *reinterpret_cast<const f32x4*>(p.resid + (size_t)orow * p.ldo + n);
                *reinterpret_cast<f32x4*>((float*)p.out + (size_t)orow * p.ldo + n) = rs + p.alpha * x;
            }
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm32_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = (p.N + 255) / 256;
    const int tiles_m = (p.M + 255) / 256;
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int GM = p.raster_gm > 0 ? p.raster_gm : 1;
    const int gsz = GM * tiles_n;
    const int g = t / gsz, r = t % gsz;
    const int gm = min(GM, tiles_m - g * GM);
    const int m0 = (g * GM + r % gm) * 256, n0 = (r / gm) * 256;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* W = (const bf16_t*)p.W;

    f32x16 acc[2][4];
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
        for (int im = 0; im < 4; ++im)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[jn][im][e] = 0.f;

    constexpr int TB = 256 * 128;     // bytes per operand tile
    const int nk = p.K / GEMM_BK;
    stage_glds32(A, p.lda, m0, 0, smem, wave, lane);
    stage_glds32(W, p.ldw, n0, 0, smem + TB, wave, lane);
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * 2 * TB;
        char* nxt = smem + ((kt + 1) & 1) * 2 * TB;
        __syncthreads();
        if (kt + 1 < nk) {
            stage_glds32(A, p.lda, m0, (kt + 1) * GEMM_BK, nxt, wave, lane);
            stage_glds32(W, p.ldw, n0, (kt + 1) * GEMM_BK, nxt + TB, wave, lane);
        }
        const char* tA = cur;
        const char* tW = cur + TB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 w[2], a[4];
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const int row = wn * 64 + jn * 32 + l31;
                w[jn] = *reinterpret_cast<const bf16x8*>(tW + row * 128 + (((ks * 2 + h) ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int im = 0; im < 4; ++im) {
                const int row = wm * 128 + im * 32 + l31;
                a[im] = *reinterpret_cast<const bf16x8*>(tA + row * 128 + (((ks * 2 + h) ^ ((row >> 1) & 7)) << 4));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int im = 0; im < 4; ++im)
                    acc[jn][im] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[jn], a[im], acc[jn][im], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    }

    const int nb = n0 + wn * 64;
#pragma unroll
    for (int im = 0; im < 4; ++im) {
        f32x4 v[8];
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                v[jn * 4 + g4] = f32x4{acc[jn][im][g4 * 4 + 0], acc[jn][im][g4 * 4 + 1], acc[jn][im][g4 * 4 + 2],
                                       acc[jn][im][g4 * 4 + 3]};
        gemm_epilogue_row32<EPI>(v, p, m0 + wm * 128 + im * 32 + l31, nb, h);
    }
}

template <int EPI>
static hipError_t launch32_t(GemmArgs a, hipStream_t s) {
    if (a.raster_gm <= 0) a.raster_gm = 4;
    const int tiles = ((a.N + 255) / 256) * ((a.M + 255) / 256);
    auto k = gemm32_bf16_kernel<EPI>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 256 * 128); attr = true; }
    hipLaunchKernelGGL(k, dim3(tiles), dim3(512), 4 * 256 * 128, s, a);
    return hipGetLastError();
}

hipError_t launch_gemm32(const GemmArgs& a, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_BF16: return launch32_t<EPI_BF16>(a, s);
        case EPI_GELU: return launch32_t<EPI_GELU>(a, s);
        case EPI_F32: return launch32_t<EPI_F32>(a, s);
        case EPI_RESID: return launch32_t<EPI_RESID>(a, s);
        case EPI_SWIGLU: return launch32_t<EPI_SWIGLU>(a, s);
        case EPI_ROPE: return launch32_t<EPI_ROPE>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vr
