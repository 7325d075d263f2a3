// SigLIP patch embedding with the patch tiles built in LDS: ToTensor + Normalize(0.5, 0.5)
// (modeling_minicpmv.py:84-92) and the Conv2d(3, D, kernel = stride = P) of timm's PatchEmbed
// (patch_embed.py:65-93) as ONE kernel — no im2col buffer in HBM.
//
//   out[(img, py, px)][n] = sum_k A[(img, py, px)][k] * Wp[n][k] + bias[n] + pos[(py, px)][n]      (fp32)
//
// K order is the image's own byte order inside a patch, k = ky * 3P + kx * 3 + c (P segments of 3P
// contiguous bytes of the HWC image per patch); the conv weight [D][3][P][P] is permuted to that order
// once at load time (pack_patch_weight).  Block tile 128 patches x 128 channels x 64 k, 4 waves, the
// MFMA loop of gemm_core.h; the W tile arrives by LDS-DMA, the A tile is written by the threads:
// u8 pixels -> (x / 255 - 0.5) / 0.5 -> bf16, 16-byte chunks at the swizzled position the fragment
// reads expect.  The pixel bytes of step t+1 are loaded before the MFMAs of step t and converted after
// them (issue early, write late).  K = 3 P^2 (588) is padded to a multiple of 64 with zeros IN LDS.
// Roofline: MFMA for the tile product; per patch 3 P^2 bytes in, D * 4 bytes out.
#include "gemm_core.h"
#include "gemm_epilogue.h"
#include "kernels.h"

namespace vr {

struct PatchArgs {
    const uint8_t* const* imgs;      // device array of n_imgs HWC uint8 images (H x W x 3)
    int H, W, P, gw, N;              // grid width, patches per image
    GemmArgs g;                      // W (permuted, bf16 [D_pad][K_pad]), bias, rowbias (pos), out, ldo, M, N, K = K_pad
    int Kreal;                       // 3 P^2
};

__global__ __launch_bounds__(256) void patch_embed_kernel(PatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const GemmArgs& p = a.g;
    const int tiles_n = p.N / GEMM_BN;
    const int tiles_m = (p.M + GEMM_BM - 1) / GEMM_BM;
    const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (t / tiles_n) * GEMM_BM, n0 = (t % tiles_n) * GEMM_BN;   // the n-tiles of an m-tile run on one XCD: pixels stay in its L2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int rowbytes = a.W * 3, seg = a.P * 3;

    // this thread's four 16-byte chunks of an A tile: rows r_i = (tid + 256 i) / 8, chunk kc = tid & 7
    const int kc = tid & 7;
    const uint8_t* pbase[4];
    bool rok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        const int m = min(m0 + row, p.M - 1);
        rok[i] = m0 + row < p.M;
        const int img = m / a.N, pp = m % a.N;
        const int py = pp / a.gw, px = pp % a.gw;
        pbase[i] = a.imgs[img] + (size_t)(py * a.P) * rowbytes + px * seg;
    }
    // byte e of chunk (k0 + kc*8): k = k0 + kc*8 + e -> image row ky = k / seg, offset k % seg
    uint8_t px8[4][8];
    auto load_pixels = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + kc * 8 + e;
            const int kk = min(k, a.Kreal - 1);
            const int off = (kk / seg) * rowbytes + (kk % seg);
#pragma unroll
            for (int i = 0; i < 4; ++i) px8[i][e] = pbase[i][off];
        }
    };
    auto write_tile = [&](char* tile, int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 3) + 32 * i;
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = ((float)px8[i][e] / 255.0f - 0.5f) / 0.5f;
                v[e] = (rok[i] && k0 + kc * 8 + e < a.Kreal) ? f2bf(f) : (bf16_t)0.f;
            }
            *reinterpret_cast<bf16x8*>(tile + row * 128 + ((kc ^ (row & 7)) << 4)) = v;
        }
    };

    gemm_acc_t acc;
    gemm_zero(acc);
    const bf16_t* Wp = (const bf16_t*)p.W;
    const int nk = p.K / GEMM_BK;
    load_pixels(0);
    stage_glds(Wp, p.ldw, n0, 0, smem + GEMM_TILE_BYTES, wave, lane);
    write_tile(smem, 0);
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * 2 * GEMM_TILE_BYTES;
        char* nxt = smem + ((kt + 1) & 1) * 2 * GEMM_TILE_BYTES;
        __syncthreads();                 // step kt complete in LDS (A written, W landed); step kt-1 consumed
        if (kt + 1 < nk) {
            load_pixels((kt + 1) * GEMM_BK);
            stage_glds(Wp, p.ldw, n0, (kt + 1) * GEMM_BK, nxt + GEMM_TILE_BYTES, wave, lane);
        }
        gemm_compute_tile(acc, cur, cur + GEMM_TILE_BYTES, wm, wn, lane);
        if (kt + 1 < nk) write_tile(nxt, (kt + 1) * GEMM_BK);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        gemm_epilogue_row<EPI_F32>(acc[i], p, m0 + wm * 64 + i * 16 + (lane & 15), n0 + wn * 64, lane >> 4);
}

hipError_t launch_patch_embed(const uint8_t* const* imgs, int n_imgs, int H, int W, int P, const GemmArgs& g, int Kreal,
                              hipStream_t s) {
    if (n_imgs <= 0) return hipSuccess;
    if (H % P || W % P || g.N % GEMM_BN || g.K % GEMM_BK || Kreal != 3 * P * P || Kreal > g.K || g.rowmap || g.ksplit > 1)
        return hipErrorInvalidValue;
    PatchArgs a{};
    a.imgs = imgs; a.H = H; a.W = W; a.P = P; a.gw = W / P; a.N = (H / P) * (W / P); a.g = g; a.Kreal = Kreal;
    const int tiles = (g.N / GEMM_BN) * ((g.M + GEMM_BM - 1) / GEMM_BM);
    static unsigned long long attr = 0;     // bit d: set on device d
    set_max_dynamic_lds((const void*)patch_embed_kernel, GEMM_SMEM_BYTES, attr);
    hipLaunchKernelGGL(patch_embed_kernel, dim3(tiles), dim3(256), GEMM_SMEM_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace vr
