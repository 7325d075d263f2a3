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
constexpr int SWEEP256_SMEM = G256_SMEM_BYTES + (int)sizeof(Sweep256Lds) + 256;   // + touch dump

__device__ __forceinline__ uint64_t ld_key(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_key(unsigned long long* p, uint64_t v) {
    __hip_atomic_store(p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}

// LDS-DMA staging of one 256 x 64 operand tile (this wave: rows [wave*32, wave*32+32)) from a
// wave-uniform base plus four per-lane 32-bit byte offsets — A and W share the offsets (same row
// pitch), so the main loop keeps 4 address VGPRs instead of 16 64-bit pointers.
__device__ __forceinline__ void stage256(const char* base, const uint32_t (&off)[4], char* tile, int wave) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        uint32_t o = off[t];
        asm volatile("" : "+v"(o));      // keep the zero-extension next to the add: saddr + 32-bit voffset form
        __builtin_amdgcn_global_load_lds(VR_GLOBAL(base + o), VR_LDS(tile + (wave * 32 + t * 8) * 128), 16, 0, 0);
    }
}

template <int KP>
__global__ __launch_bounds__(512) void search_sweep256_kernel(SearchArgs p, int q_tiles, int tiles_per_chunk,
                                                              const float* __restrict__ thr_init) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Sweep256Lds& L = *reinterpret_cast<Sweep256Lds*>(smem + G256_SMEM_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fq = lane >> 4;

    // (the runtime division runs on the VALU: pin the wave-uniform results back into SGPRs)
    const int b = blockIdx.x;
    const int chunk = __builtin_amdgcn_readfirstlane((b / (8 * q_tiles)) * 8 + (b & 7));   // the query tiles of a chunk share an XCD
    const int qt = __builtin_amdgcn_readfirstlane((b >> 3) % q_tiles);
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

    // one wave compacts the buffers of its 32 queries: sort, keep the best KP, raise thr.
    // Four queries per round: the four key loads (L2 round trips) fly together and the four
    // independent sort networks interleave on the VALU.
    auto compact = [&](bool force, int wave, int lane) {
        const int c_l = L.cnt[wave * 32 + (lane & 31)];
        unsigned long long todo = __ballot((lane < 32) && (force ? c_l > 0 : c_l > SRCH_TRIG));
        while (todo) {
            int qq[4], c[4];
            uint64_t key[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int src = todo ? __ffsll((long long)todo) - 1 : -1;
                todo &= todo - 1;                                   // (0 stays 0)
                qq[u] = src < 0 ? -1 : wave * 32 + src;
                c[u] = src < 0 ? 0 : __shfl(c_l, src, 64);
                key[u] = (lane < c[u]) ? ld_key(gc + (size_t)qq[u] * gq + lane) : KEY_NONE;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) key[u] = wave_sort_desc(key[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (qq[u] < 0) continue;                            // wave-uniform
                if (lane < KP) st_key(gc + (size_t)qq[u] * gq + lane, key[u]);
                if (lane == KP - 1 && c[u] >= KP) L.thr[qq[u]] = orderable_f32((uint32_t)(key[u] >> 32));
                if (lane == 0) L.cnt[qq[u]] = min(c[u], KP);
            }
        }
    };

    // ---- one K-step stream over all tiles of the chunk -----------------------------------------
    // The index rows come from HBM, not from L2: with two LDS stages the LDS-DMA of step s+1 has
    // one step (~1.1 us) to land, less than an HBM round trip under load, and every K-step stalled
    // (measured: 680 TFLOP/s against 1.2 PFLOP/s for the same loop on L2-resident operands).
    // Each step therefore also TOUCHES the index lines of step s+3 (one dword per thread, 2 per
    // 128-B line, LDS-DMA'd into a dump area so no register is written): the line is in L2 when
    // the real DMA asks for it two steps later.  The touch is the youngest VMEM op of the step, so
    // `vmcnt(1)` waits for the stage without waiting for the touch.  The DMA of the next tile's
    // first step is issued before the filter epilogue of the current tile.
    const char* A = (const char*)p.index_bf16;
    const char* W = (const char*)p.q_bf16 + (size_t)q0 * p.dim * 2;
    const int nk = p.dim / GEMM_BK;
    const size_t tile_bytes = (size_t)256 * p.dim * 2;       // one 256-row tile of the index
    constexpr int SB = 2 * G256_TILE_BYTES;
    char* dump = smem + G256_SMEM_BYTES + sizeof(Sweep256Lds);
    const int arow = wm * 128 + fr, wrow = wn * 64 + fr;
    uint32_t off[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = wave * 32 + t * 8 + (lane >> 3);
        off[t] = (uint32_t)(row * p.dim + (((lane & 7) ^ (row & 7)) << 3)) * 2u;
    }
    const uint32_t toff = (uint32_t)((tid >> 1) * p.dim + (tid & 1) * 32) * 2u;
    auto touch = [&](const char* line) {                     // line: wave-uniform address of row 0's 128-B line
        uint32_t o = toff;
        asm volatile("" : "+v"(o));
        __builtin_amdgcn_global_load_lds(VR_GLOBAL(uniform_ptr(line) + o), VR_LDS(dump), 4, 0, 0);
    };
    int sp = 0;
    if (tile_lo < tile_hi) {
        touch(A + (size_t)tile_lo * tile_bytes + min(1, nk - 1) * (GEMM_BK * 2));
        touch(A + (size_t)tile_lo * tile_bytes + min(2, nk - 1) * (GEMM_BK * 2));
        stage256(A + (size_t)tile_lo * tile_bytes, off, smem, wave);
        stage256(W, off, smem + G256_TILE_BYTES, wave);
    }
    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        const int doc0 = tile * 256;
        const char* At = A + (size_t)tile * tile_bytes;
        gemm256_acc_t acc;
        gemm256_zero(acc);
        for (int kt = 0; kt < nk; ++kt) {
            char* cur = smem + sp * SB;
            char* nxt = smem + (sp ^ 1) * SB;
            if (kt == 0) VR_WAIT_VM_BARRIER(0);              // (epilogue stores may sit behind the touch)
            else VR_WAIT_VM_BARRIER(1);
            const bool wrap = kt + 1 == nk;                  // next step opens the next tile
            if (!wrap || tile + 1 < tile_hi) {
                const char* nA = uniform_ptr(wrap ? At + tile_bytes : At + (kt + 1) * (GEMM_BK * 2));
                const char* nW = uniform_ptr(wrap ? W : W + (kt + 1) * (GEMM_BK * 2));
                stage256(nA, off, nxt, wave);
                stage256(nW, off, nxt + G256_TILE_BYTES, wave);
            }
            // step s+3: in this tile, in the next one, or (nothing left) the current line again
            const int k3 = kt + 3;
            touch(k3 < nk ? At + k3 * (GEMM_BK * 2)
                          : (tile + 1 < tile_hi ? At + tile_bytes + min(k3 - nk, nk - 1) * (GEMM_BK * 2)
                                                : At + kt * (GEMM_BK * 2)));
            gemm256_compute_il(acc, cur, cur + G256_TILE_BYTES, arow, wrow, fq);
            sp ^= 1;
        }
        // ---- filter epilogue.  Everything it derives from the lane / wave id is recomputed here
        // from laundered copies: hoisted above the K-loop those values spilled the main loop.
        int lane_e = lane, wave_e = wave;
        asm volatile("" : "+v"(lane_e), "+s"(wave_e));
        const int fr_e = lane_e & 15, fq_e = lane_e >> 4, wm_e = wave_e >> 2, wn_e = wave_e & 3;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int doc = doc0 + wm_e * 128 + i * 16 + fr_e;
            bool any = false;
            if (doc < p.n_docs) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int qn = wn_e * 64 + j * 16 + fq_e * 4;
                    const f32x4 th = *reinterpret_cast<const f32x4*>(&L.thr[qn]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float s = acc[i][j][r];
                        if (s >= th[r]) {
                            const int qq = qn + r;
                            // LDS counter bump in asm: as a builtin the compiler puts s_waitcnt vmcnt(0)
                            // in front of it (LDS-DMA alias rule), i.e. every append waited for the
                            // previous append's global store and for the next tile's in-flight stage
                            int pos;
                            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                                         : "=v"(pos) : "v"((uint32_t)(uintptr_t)VR_LDS(&L.cnt[qq])), "v"(1) : "memory");
                            st_key(gc + (size_t)qq * gq + pos, make_key(s, (uint32_t)doc));
                            any = true;
                        }
                    }
                }
            }
            // a query gains at most 2 x 16 candidates per strip, so TRIG + 32 <= CAP never overflows
            if (__syncthreads_or(any)) {
                compact(false, wave_e, lane_e);
                __syncthreads();
            }
        }
    }
    __syncthreads();
    compact(true, wave, lane);
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
