// Search sweep on the 256 x 256 x 64 tile (8 waves, 128 KiB of LDS stages) + its merge kernel:
// the main sweep of vr_index_search for more than 128 queries.
//
// Same idea as search_sweep_kernel (search.hip) — fused similarity GEMM, scores filtered in
// registers against a per-query threshold, survivors kept as (score, id) keys, exact top-k by a
// 64-lane bitonic network — on the GEMM main loop that reaches ~1.1 PFLOP/s here (the 128^2 loop
// with 64 KiB of LDS candidate buffers ran one 4-wave workgroup per CU: 390 TFLOP/s).
//
//   workgroup = (doc chunk, 256-query tile); A = index rows (docs), W = query rows.
//   acc[i][j][r]: doc = wm*128 + i*16 + fr, query = wn*64 + j*16 + fq*4 + r.
//
// Candidate lists.  The LDS is taken by the operand stages, so survivors go to a global scratch
// (L2-resident; ~30 appends per (query, chunk) after the threshold pre-pass).  Every wave OWNS the
// half-lists [query][chunk][wm][64] of its 64 queries: the 16 lanes of a quad-row group (same fq)
// hold the same 16 query columns, so a wave-wide ballot gives every lane of the group both its
// slot (popcount of the group's lower lanes) and the group's new length — the list lengths live
// replicated in registers (16 byte-counters per lane), no atomics, no LDS traffic, no barriers in
// the epilogue.  A half-list that passes 48 entries (adversarial orders; ~never after the
// pre-pass) is compacted by its wave alone: sort, keep KP, raise the query's threshold.
// The lists are emitted UNSORTED with their lengths; search_merge256_kernel bounds the global
// KP-th best from lane-local maxima, filters, sorts the few survivors once and re-scores in fp32.
#include <cstdlib>

#include "gemm_core.h"
#include "gemm_core_il.h"
#include "kernels.h"
#include "search_common.h"

namespace vr {

constexpr int HL_CAP = 64;          // slots per half-list
constexpr int HL_TRIG = 48;         // compact a half-list longer than this (a strip adds <= 16)
constexpr int HL_COMPACTED = 0x100; // flag in a half-list's emitted length: it was compacted (rows were dropped from it)

struct Sweep256Lds {
    float thr[256];
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
                                                              const float* __restrict__ thr_init, int debug) {
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
    // half-list of query qq (0..255 in this tile), owner half wm: gw + qq * gq
    unsigned long long* gw = p.cand_keys + (((size_t)q0 * p.n_chunks + chunk) * 2 + wm) * HL_CAP;
    const size_t gq = (size_t)p.n_chunks * 2 * HL_CAP;

    if (tid < 256) {
        float t0 = (thr_init && q0 + tid < p.nq) ? thr_init[q0 + tid] : -INFINITY;
        if (q0 + tid >= p.nq || (debug & 1)) t0 = INFINITY;    // padding queries never collect candidates
        L.thr[tid] = t0;
    }
    __syncthreads();

    // lengths of this wave's half-lists, replicated over the 16 lanes of a group:
    // byte r of c8[j] = length of the list of query wn*64 + j*16 + fq*4 + r
    uint32_t c8[4] = {0u, 0u, 0u, 0u};
    uint32_t cmask = 0u;                 // bit j*4 + r: that half-list was compacted (merge: rows may be missing from it)

    // rare: a half-list of this wave passed HL_TRIG -> the wave alone sorts it, keeps the best KP and
    // raises the query's threshold.  A runtime loop over the 16 (j, r) columns (kept rolled: unrolled
    // into the 8 strips it bloated the epilogue until the strip loop no longer unrolled).
    auto compact_own = [&](int lane_e, int wave_e) {
        const int fr_e = lane_e & 15, fq_e = lane_e >> 4, wn_e = wave_e & 3;
        __threadfence_block();                                  // own stores visible to own loads
#pragma nounroll
        for (int jr = 0; jr < 16; ++jr) {
            const int j = jr >> 2, r = jr & 3;
            const uint32_t cw = j == 0 ? c8[0] : j == 1 ? c8[1] : j == 2 ? c8[2] : c8[3];
            const int c_l = (int)((cw >> (8 * r)) & 0xFFu);
            unsigned long long todo = __ballot(c_l > HL_TRIG && fr_e == 0);   // one bit per group
            uint32_t nw = cw;
            while (todo) {
                const int src = __ffsll((long long)todo) - 1;                // lane fq*16
                todo &= todo - 1;
                const int qq = wn_e * 64 + j * 16 + (src >> 4) * 4 + r;
                const int c = __shfl(c_l, src, 64);
                unsigned long long* row = gw + (size_t)qq * gq;
                uint64_t key = (lane_e < c) ? ld_key(row + lane_e) : KEY_NONE;
                key = wave_sort_desc(key);
                if (lane_e < KP) st_key(row + lane_e, key);
                const int keep = min(c, KP);
                if (lane_e == KP - 1 && c >= KP) {
                    const float t = orderable_f32((uint32_t)(key >> 32));
                    if (t > L.thr[qq]) L.thr[qq] = t;              // (the partner half may race: both bounds are valid)
                }
                if (fq_e == (src >> 4)) { nw = (nw & ~(0xFFu << (8 * r))) | ((uint32_t)keep << (8 * r)); cmask |= 1u << jr; }
            }
            c8[0] = j == 0 ? nw : c8[0];
            c8[1] = j == 1 ? nw : c8[1];
            c8[2] = j == 2 ? nw : c8[2];
            c8[3] = j == 3 ? nw : c8[3];
        }
        __threadfence_block();
    };

    // ---- one K-step stream over all tiles of the chunk -----------------------------------------
    // Each step also TOUCHES the index lines of step s+3 (one dword per thread, 2 per 128-B line,
    // LDS-DMA'd into a dump area so no register is written), so the index rows — which come from
    // HBM, not from L2 — are on their way two steps before the real DMA asks for them.  The touch
    // is the youngest VMEM op of the step, so `vmcnt(1)` waits for the stage without waiting for
    // the touch.  The DMA of the next tile's first step is issued before the filter epilogue of
    // the current tile.
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
        if (debug & 2) line = A;                               // tuning aid: touch a fixed line (no prefetch effect)
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
        // ---- filter epilogue (no workgroup barrier).  Everything it derives from the lane / wave
        // id is recomputed here from laundered copies: hoisted above the K-loop those values
        // spilled the main loop.
        int lane_e = lane, wave_e = wave;
        asm volatile("" : "+v"(lane_e), "+s"(wave_e));
        const int fr_e = lane_e & 15, fq_e = lane_e >> 4, wm_e = wave_e >> 2, wn_e = wave_e & 3;
        const uint32_t below = (1u << fr_e) - 1u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int doc = doc0 + wm_e * 128 + i * 16 + fr_e;
            const bool valid = doc < p.n_docs;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int qn = wn_e * 64 + j * 16 + fq_e * 4;
                const f32x4 th = *reinterpret_cast<const f32x4*>(&L.thr[qn]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s = acc[i][j][r];
                    const bool pass = valid && s >= th[r];
                    const unsigned long long bal = __ballot(pass);
                    if (bal) {                                  // wave-uniform, ~70 % of the (j, r)
                        const uint32_t grp = (uint32_t)(bal >> (fq_e * 16)) & 0xFFFFu;
                        if (pass) {
                            const int pos = (int)((c8[j] >> (8 * r)) & 0xFFu) + __popc(grp & below);
                            st_key(gw + (size_t)(qn + r) * gq + pos, make_key(s, (uint32_t)doc));
                        }
                        c8[j] += (uint32_t)__popc(grp) << (8 * r);
                    }
                }
            }
            // ---- rare: one of this wave's half-lists passed HL_TRIG -> the wave compacts it
            bool over = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) over |= ((c8[j] + 0x01010101u * (127 - HL_TRIG)) & 0x80808080u) != 0u;
            if (__any(over)) compact_own(lane_e, wave_e);
        }
    }
    // ---- list lengths: [query][chunk][wm]
    if (fr == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = wn * 64 + j * 16 + fq * 4 + r;
                p.cand_ids[((size_t)(q0 + qq) * p.n_chunks + chunk) * 2 + wm] =
                    (int)((c8[j] >> (8 * r)) & 0xFFu) | (((cmask >> (j * 4 + r)) & 1u) ? HL_COMPACTED : 0);
            }
        }
    }
}


// ---- merge: one WORKGROUP per query (256 threads) over its n_chunks * 2 unsorted half-lists ---
//  1. thread t takes the max over its own lists (t, t + 256, ...): maxima over DISJOINT entry
//     sets; the KP-th largest of them is <= KP distinct entries, i.e. a lower bound of the
//     query's global KP-th best key (KEY_NONE when fewer than KP threads saw a key);
//  2. entries >= that bound are appended to LDS (ballot compaction) — typically ~20 of ~1900;
//  3. one sort (or, for > MERGE_CAP survivors = massive ties, a merge over everything);
//  4. fp32 re-scoring by the four waves concurrently, certification, final sort (certify_tail).
// All list walks of a query run in parallel: a wave-per-query merge left most of the chip idle
// behind a chain of dependent loads with a few hundred queries (128 queries: 101 us).
template <int KP>
__global__ __launch_bounds__(256) void search_merge256_wg_kernel(SearchArgs p) {
    constexpr int GD = KP + MERGE_GD_EXTRA < 64 ? KP + MERGE_GD_EXTRA : 64;   // gather depth: certification may want candidates past the KP-th
    __shared__ uint64_t lm[256];
    __shared__ uint64_t surv[MERGE_CAP], exact_w[MERGE_CAP];
    __shared__ uint64_t cand[64], exact_s[64];
    __shared__ uint64_t thr_s;
    __shared__ f32x4 q_s[MERGE_MAXV * 64];               // the query for certify_tail's paired re-scoring (10 KiB)
    __shared__ int n_s, x_s, comp_s;
    __shared__ unsigned drop_s;
    __shared__ float tau_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x;
    const int lists = p.n_chunks * 2;
    const int* cnts = p.cand_ids + (size_t)q * lists;      // length | HL_COMPACTED
    const unsigned long long* keys = p.cand_keys + (size_t)q * lists * HL_CAP;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

    uint64_t m = KEY_NONE;
    int comp = 0;
    unsigned drop = 0u;                                  // orderable score: the best KP-th entry of a compacted half-list
    if (tid == 0) { comp_s = 0; drop_s = 0u; }
    for (int l = tid; l < lists; l += 256) {
        const int cw = cnts[l];
        const int c = cw & 0xFF;
        comp |= cw >> 8;
        // a compacted half-list keeps its best KP sorted in slots 0 .. KP-1 (later rows are appended behind them): slot KP-1 is
        // the threshold that compaction raised — what it dropped, and what the raised threshold kept out of this list and its
        // partner half afterwards, scores no higher.  (A half-list flagged for its partner's compaction: any entry of it, looser.)
        if ((cw >> 8) && c >= KP) drop = max(drop, (unsigned)(keys[(size_t)l * HL_CAP + KP - 1] >> 32));
        const u64x2* row = reinterpret_cast<const u64x2*>(keys + (size_t)l * HL_CAP);
        for (int e = 0; e < c; e += 4) {
            const u64x2 a = row[e >> 1], b2 = row[(e >> 1) + 1];
            const uint64_t k0 = a[0], k1 = e + 1 < c ? a[1] : KEY_NONE;
            const uint64_t k2 = e + 2 < c ? b2[0] : KEY_NONE, k3 = e + 3 < c ? b2[1] : KEY_NONE;
            const uint64_t x = k0 > k1 ? k0 : k1, y = k2 > k3 ? k2 : k3;
            const uint64_t z = x > y ? x : y;
            m = z > m ? z : m;
        }
    }
    lm[tid] = m;
    __syncthreads();
    if (comp) atomicOr(&comp_s, 1);
    if (drop) atomicMax(&drop_s, drop);
    if (wave == 0) {
        uint64_t v = lm[lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) { const uint64_t o = lm[lane + 64 * w]; v = o > v ? o : v; }
        const uint64_t t = shfl_u64(wave_sort_desc(v), GD - 1);
        if (lane == 0) thr_s = t;
    }
    __syncthreads();
    // every entry with key >= thr -> surv (<= MERGE_CAP of them), the best 64 sorted into cand; returns their number
    auto gather = [&](uint64_t thr) -> int {
        if (tid == 0) n_s = 0;
        __syncthreads();
        for (int l = tid; l < lists; l += 256) {
            const int c = cnts[l] & 0xFF;
            const u64x2* row = reinterpret_cast<const u64x2*>(keys + (size_t)l * HL_CAP);
            for (int e = 0; e < c; e += 2) {
                const u64x2 a = row[e >> 1];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (e + u < c && a[u] >= thr && a[u] != KEY_NONE) {
                        const int pos = atomicAdd(&n_s, 1);
                        if (pos < MERGE_CAP) surv[pos] = a[u];
                    }
                }
            }
        }
        __syncthreads();
        const int n = n_s;
        if (wave == 0) {
            uint64_t best = KEY_NONE;
            if (n <= MERGE_CAP) {
                for (int base = 0; base < n; base += 64) {
                    const uint64_t key = (base + lane < n) ? surv[base + lane] : KEY_NONE;
                    best = (base == 0) ? wave_sort_desc(key) : wave_merge_top64(best, key, lane);
                }
            } else {
                for (int l = 0; l < lists; ++l) {
                    const int c = cnts[l] & 0xFF;
                    const uint64_t key = (lane < c) ? keys[(size_t)l * HL_CAP + lane] : KEY_NONE;
                    best = (l == 0) ? wave_sort_desc(key) : wave_merge_top64(best, key, lane);
                }
            }
            cand[lane] = best;
        }
        __syncthreads();
        return n;
    };
    const uint64_t thr = thr_s;
    const int n = gather(thr);
    // what the re-scoring does not see (search_common.h: certify_tail):
    //   list entries outside `cand`: below the gather bound, or (more than 64 gathered) below cand[63];
    //   rows that never reached a list: below the sweep's starting threshold — or, where a half-list was compacted
    //   (its chunk's threshold rose to that list's KP-th best), below the largest such KP-th best (drop_s) and below the
    //   global KP-th best cand[KP - 1]: two valid bounds, the smaller one is taken.  (Until round 4 only the global one:
    //   on a shard too small for the pre-pass EVERY half-list is compacted, the global KP-th best lies inside the error band
    //   of the k-th, and four queries in five went to the band pass.)
    const float coverB = n > 64 ? key_score(cand[63]) : (thr == KEY_NONE ? -INFINITY : key_score(thr));
    float dropB = p.thr_used ? p.thr_used[q] : -INFINITY;
    if (comp_s && cand[KP - 1] != KEY_NONE) {
        float b = key_score(cand[KP - 1]);
        if (drop_s) b = fminf(b, orderable_f32(drop_s));
        dropB = fmaxf(dropB, b);
    }
    certify_tail<KP>(p, q, cand, exact_s, coverB, dropB, &tau_s, &x_s,
                     [&](float tau) { return gather((uint64_t)f32_orderable(tau) << 32); }, surv, exact_w, false, q_s);
}

template <int KP>
static hipError_t launch_t(const SearchArgs& a, const float* thr, hipStream_t s) {
    const int q_tiles = (a.nq + 255) / 256;
    const int n_tiles = (int)((a.n_docs + 255) / 256);
    const int tpc = (n_tiles + a.n_chunks - 1) / a.n_chunks;
#ifndef VR_SWEEP_W
#define VR_SWEEP_W 1
#endif
    hipError_t e;
    if (a.pre_own_chunks && !(VR_SWEEP_W && sweep256w_ok(a))) return hipErrorInvalidValue;   // (only that sweep skips tiles)
    if (VR_SWEEP_W && sweep256w_ok(a)) {     // the one-wave-per-SIMD form of the sweep (search256w.hip)
        e = launch_sweep256w(a, KP, thr, s);
    } else {
        auto k = search_sweep256_kernel<KP>;
        static unsigned long long attr = 0;     // bit d: set on device d
        set_max_dynamic_lds((const void*)k, SWEEP256_SMEM, attr);
        hipLaunchKernelGGL(k, dim3(a.n_chunks * q_tiles), dim3(512), SWEEP256_SMEM, s, a, q_tiles, tpc, thr, 0);
        e = hipGetLastError();
    }
    if (e != hipSuccess) return e;
    if (a.prof_ev && (e = hipEventRecord(a.prof_ev[3], s)) != hipSuccess) return e;
    hipLaunchKernelGGL(search_merge256_wg_kernel<KP>, dim3(a.nq), dim3(256), 0, s, a);   // one workgroup per query
    return hipGetLastError();
}

// sweep + merge of the 256-tile path; cand_keys: [nq_pad256][n_chunks][2][64] keys,
// cand_ids: [nq_pad256][n_chunks][2] list lengths
hipError_t launch_sweep256(const SearchArgs& a, int kp, const float* thr, hipStream_t s) {
    if (!a.cand_keys || a.n_chunks % 8) return hipErrorInvalidValue;
    switch (kp) {
        case 16: return launch_t<16>(a, thr, s);
        case 32: return launch_t<32>(a, thr, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vr
