// Search sweep on the 256 x 256 x 64 tile (8 waves, 128 KiB of LDS stages): the main sweep of
// vr_index_search for more than 128 queries.
//
// Same algorithm as search_sweep_kernel (search.hip) — fused similarity GEMM, per-query running
// threshold, 64-entry candidate buffers compacted by a 64-lane bitonic sort — but on the GEMM
// main loop that reaches ~1.2 PFLOP/s instead of the 128^2 loop that, with 64 KiB of candidate
// buffers next to its stages, ran one 4-wave workgroup per CU (390 TFLOP/s).  The LDS is all
// taken by the operand stages, so the candidate buffers live in a global scratch
// ([query][chunk][64] keys, L2-resident: ~30 appends per (query, chunk) after the threshold
// pre-pass); only thresholds and counters stay in LDS.
//   workgroup = (doc chunk, 256-query tile); A = index rows (docs), W = query rows.
//   acc[i][j][r]: doc = wm*128 + i*16 + fr, query = wn*64 + j*16 + fq*4 + r.
#include "gemm_core.h"
#include "gemm_core_il.h"
#include "kernels.h"
#include "search_common.h"

namespace vr {

struct Sweep256Lds {
    float thr[256];
    int cnt[256];
};
constexpr int SWEEP256_SMEM = G256_SMEM_BYTES + (int)sizeof(Sweep256Lds);

__device__ __forceinline__ uint64_t ld_key(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_key(unsigned long long* p, uint64_t v) {
    __hip_atomic_store(p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int KP>
__global__ __launch_bounds__(512) void search_sweep256_kernel(SearchArgs p, int q_tiles, int tiles_per_chunk,
                                                              const float* __restrict__ thr_init) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Sweep256Lds& L = *reinterpret_cast<Sweep256Lds*>(smem + G256_SMEM_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fq = lane >> 4;

    const int b = blockIdx.x;
    const int chunk = (b / (8 * q_tiles)) * 8 + (b & 7);       // the query tiles of a chunk share an XCD
    const int qt = (b >> 3) % q_tiles;
    const int q0 = qt * 256;
    const int n_tiles = (int)((p.n_docs + 255) / 256);
    const int tile_lo = chunk * tiles_per_chunk;
    const int tile_hi = min(n_tiles, tile_lo + tiles_per_chunk);
    unsigned long long* gc = p.cand_keys + ((size_t)q0 * p.n_chunks + chunk) * SRCH_CAP;
    const size_t gq = (size_t)p.n_chunks * SRCH_CAP;           // stride between queries

    if (tid < 256) {
        float t0 = (thr_init && q0 + tid < p.nq) ? thr_init[q0 + tid] : -INFINITY;
        if (q0 + tid >= p.nq) t0 = INFINITY;                   // padding queries never collect candidates
        L.thr[tid] = t0;
        L.cnt[tid] = 0;
    }
    __syncthreads();

    // one wave compacts the buffers of its 32 queries: sort, keep the best KP, raise thr
    auto compact = [&](bool force) {
        const int c_l = L.cnt[wave * 32 + (lane & 31)];
        unsigned long long todo = __ballot((lane < 32) && (force ? c_l > 0 : c_l > SRCH_TRIG));
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int qq = wave * 32 + src;
            const int c = __shfl(c_l, src, 64);
            unsigned long long* row = gc + (size_t)qq * gq;
            uint64_t key = (lane < c) ? ld_key(row + lane) : KEY_NONE;
            key = wave_bitonic_desc(key, lane);
            if (lane < KP) st_key(row + lane, key);
            if (lane == KP - 1 && c >= KP) L.thr[qq] = orderable_f32((uint32_t)(key >> 32));
            if (lane == 0) L.cnt[qq] = min(c, KP);
        }
    };

    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        const int doc0 = tile * 256;
        gemm256_acc_t acc;
        gemm256_zero(acc);
        gemm256_mainloop_il(acc, (const bf16_t*)p.index_bf16, p.dim, (const bf16_t*)p.q_bf16, p.dim, doc0, q0,
                            p.dim, smem);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int doc = doc0 + wm * 128 + i * 16 + fr;
            bool any = false;
            if (doc < p.n_docs) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int qn = wn * 64 + j * 16 + fq * 4;
                    const f32x4 th = *reinterpret_cast<const f32x4*>(&L.thr[qn]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float s = acc[i][j][r];
                        if (s >= th[r]) {
                            const int pos = atomicAdd(&L.cnt[qn + r], 1);
                            st_key(gc + (size_t)(qn + r) * gq + pos, make_key(s, (uint32_t)doc));
                            any = true;
                        }
                    }
                }
            }
            // a query gains at most 2 x 16 candidates per strip, so TRIG + 32 <= CAP never overflows
            if (__syncthreads_or(any)) {
                compact(false);
                __syncthreads();
            }
        }
    }
    __syncthreads();
    compact(true);
    __syncthreads();
    // emit [query][chunk][KP] (score, id); unused slots: -inf / -1
    for (int e = tid; e < 256 * KP; e += 512) {
        const int qq = e / KP, s = e % KP;
        const size_t o = ((size_t)(q0 + qq) * p.n_chunks + chunk) * KP + s;
        if (s < L.cnt[qq]) {
            const uint64_t key = ld_key(gc + (size_t)qq * gq + s);
            p.cand_scores[o] = orderable_f32((uint32_t)(key >> 32));
            p.cand_ids[o] = (int)(~(uint32_t)key);
        } else {
            p.cand_scores[o] = -INFINITY;
            p.cand_ids[o] = -1;
        }
    }
}

template <int KP>
static hipError_t launch_t(const SearchArgs& a, const float* thr, hipStream_t s) {
    const int q_tiles = (a.nq + 255) / 256;
    const int n_tiles = (int)((a.n_docs + 255) / 256);
    const int tpc = (n_tiles + a.n_chunks - 1) / a.n_chunks;
    auto k = search_sweep256_kernel<KP>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, SWEEP256_SMEM); attr = true; }
    hipLaunchKernelGGL(k, dim3(a.n_chunks * q_tiles), dim3(512), SWEEP256_SMEM, s, a, q_tiles, tpc, thr);
    return hipGetLastError();
}

hipError_t launch_sweep256(const SearchArgs& a, int kp, const float* thr, hipStream_t s) {
    if (!a.cand_keys || a.n_chunks % 8) return hipErrorInvalidValue;
    switch (kp) {
        case 16: return launch_t<16>(a, thr, s);
        case 32: return launch_t<32>(a, thr, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vr
