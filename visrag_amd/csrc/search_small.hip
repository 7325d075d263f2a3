// vr_index_search for a HANDFUL of queries (nq <= 16): the HBM-bound regime of the roofline
// (bytes = index size).  The fused sweeps run the GEMM main loop with 7/8 of the
// query tile empty and reach ~1.5 TB/s of index reads; here the index is simply STREAMED:
//   * one 512-thread workgroup per CU owns a contiguous row range; its 8 waves take 16-row strips;
//   * the (<= 16) queries sit in LDS once, pre-arranged as MFMA B fragments; an index strip goes
//     global -> registers -> v_mfma_f32_16x16x32_bf16 with no LDS stage: a lane loads 32
//     contiguous bytes of its row per K pair-step (4 lanes = one full 128-B line), the k-index
//     permutation this implies is applied to the query fragments instead;
//   * loads run a chunk (4 pair-steps = 8 KiB per wave) ahead of the MFMAs, across strip borders;
//   * lane (query = fr) filters its 4 scores per strip against the query's running threshold and
//     appends survivors to a per-query LDS list (192 slots: a round of 8 strips adds <= 128);
//     after every round the lists are compacted to the best KP by the 64-lane sorting network.
// Output: sorted per-(query, workgroup) lists in the chunk-list format of search_merge_wg_kernel.
#include <cstdlib>

#include "kernels.h"
#include "search_common.h"

namespace vr {

constexpr int SS_WG = 256;           // workgroups = chunk lists per query
constexpr int SS_CAP = 192;          // list slots per query
constexpr int SS_CH = 4;             // pair-steps per prefetch chunk

int search_stream_chunks() { return SS_WG; }

bool search_uses_stream(int nq, int dim) {
    return nq <= 16 && dim % 256 == 0 && dim <= 2560;
}

struct StreamLds {
    float thr[16];
    int cnt[16];
    uint64_t cand[16][SS_CAP];
};

template <int KP>
__global__ __launch_bounds__(512) void search_stream_kernel(SearchArgs p, int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int dim = p.dim;
    const int nps = dim >> 6;                                  // pair-steps (64 k each)
    char* qs = smem;                                           // [2*nps][64 lanes][16 B]
    StreamLds& L = *reinterpret_cast<StreamLds*>(smem + (size_t)dim * 32);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;

    // ---- queries -> LDS in B-fragment order: step 2t+h, lane (fq, fr) <- q[fr][t*64 + fq*16 + h*8 .. +7]
    for (int c = tid; c < 2 * nps * 64; c += 512) {
        const int step = c >> 6, l = c & 63;
        const int t = step >> 1, h = step & 1, qf = l & 15, qq = l >> 4;
        const size_t o = (size_t)qf * dim + t * 64 + qq * 16 + h * 8;
        if (p.convert_q) {           // straight from the caller's fp32 rows (the same rounding as launch_f32_to_bf16): no convert launch
            bf16x8 v;
            if (qf < p.nq) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(p.q_f32 + o), b = *reinterpret_cast<const f32x4*>(p.q_f32 + o + 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = f2bf(a[r]); v[4 + r] = f2bf(b[r]); }
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = f2bf(0.f);
            }
            *reinterpret_cast<bf16x8*>(qs + (size_t)c * 16) = v;
        } else {
            *reinterpret_cast<u32x4*>(qs + (size_t)c * 16) = *reinterpret_cast<const u32x4*>((const bf16_t*)p.q_bf16 + o);
        }
    }
    if (p.convert_q && blockIdx.x == 0) {       // what the convert launch left behind for the merge / band pass
        bf16_t* qb = (bf16_t*)const_cast<void*>(p.q_bf16);
        for (int i = tid; i < p.nq * dim / 8; i += 512) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p.q_f32 + (size_t)i * 8), b = *reinterpret_cast<const f32x4*>(p.q_f32 + (size_t)i * 8 + 4);
            bf16x8 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = f2bf(a[r]); v[4 + r] = f2bf(b[r]); }
            *reinterpret_cast<bf16x8*>(qb + (size_t)i * 8) = v;
        }
        if (tid == 0) {
            if (p.flag_count) p.flag_count[0] = 0;
            if (p.flag2_count) p.flag2_count[0] = 0;
        }
    }
    if (tid < 16) { L.thr[tid] = tid < p.nq ? -INFINITY : INFINITY; L.cnt[tid] = 0; }
    __syncthreads();

    const int64_t row_lo = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t row_hi = min(p.n_docs, row_lo + rows_per_wg);
    const int n_strips = row_hi > row_lo ? (int)((row_hi - row_lo + 15) / 16) : 0;
    const int rounds = (n_strips + 7) / 8;
    const char* ibase = (const char*)p.index_bf16;
    const size_t row_bytes = (size_t)dim * 2;
    const int n_ch = nps / SS_CH;                              // chunks per strip (dim % 256 == 0)

    // this lane's row of a strip (clamped to the last valid row: masked later), byte offset in the 128-B line
    auto row_ptr = [&](int strip) -> const char* {
        const int64_t r = min(row_lo + (int64_t)strip * 16 + fr, p.n_docs - 1);
        return ibase + (size_t)r * row_bytes + fq * 32;
    };
    u32x4 bufA[2 * SS_CH], bufB[2 * SS_CH];
    auto load_chunk = [&](u32x4 (&b)[2 * SS_CH], const char* rp, int ch) {
#pragma unroll
        for (int u = 0; u < SS_CH; ++u) {
            const char* a = rp + (size_t)(ch * SS_CH + u) * 128;
            b[2 * u] = *reinterpret_cast<const u32x4*>(a);
            b[2 * u + 1] = *reinterpret_cast<const u32x4*>(a + 16);
        }
    };
    auto mma_chunk = [&](f32x4& acc, const u32x4 (&b)[2 * SS_CH], int ch) {
#pragma unroll
        for (int u = 0; u < 2 * SS_CH; ++u) {
            const bf16x8 qb = *reinterpret_cast<const bf16x8*>(qs + ((size_t)(ch * 2 * SS_CH + u) * 64 + lane) * 16);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[u]), qb, acc, 0, 0, 0);
        }
    };

    for (int rd = 0; rd < rounds; ++rd) {
        const int strip = rd * 8 + wave;
        if (strip < n_strips) {
            const char* rp = row_ptr(strip);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            load_chunk(bufA, rp, 0);
            for (int ch = 0; ch < n_ch; ch += 2) {                 // n_ch may be odd: second half guarded
                if (ch + 1 < n_ch) load_chunk(bufB, rp, ch + 1);
                mma_chunk(acc, bufA, ch);
                if (ch + 1 < n_ch) {
                    if (ch + 2 < n_ch) load_chunk(bufA, rp, ch + 2);
                    mma_chunk(acc, bufB, ch + 1);
                }
            }
            // D[doc = fq*4 + r][query = fr]
            // every score leaves for the band pass a flagged query's merge workgroup may need (search_band.h): 16 bytes per lane,
            // 6.4 MB per search next to the 460 MB read (the rows of a strip past n_docs repeat the last row: never looked at)
            if (p.score_rows && fr < p.nq)
                *reinterpret_cast<f32x4*>(p.score_rows + (size_t)fr * p.ld_scores + (size_t)(row_lo + (int64_t)strip * 16 + fq * 4)) = acc;
            if (fr < p.nq) {
                const float th = L.thr[fr];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t doc = row_lo + (int64_t)strip * 16 + fq * 4 + r;
                    if (doc < row_hi && acc[r] >= th) {
                        const int pos = atomicAdd(&L.cnt[fr], 1);
                        L.cand[fr][pos] = make_key(acc[r], (uint32_t)doc);
                    }
                }
            }
        }
        __syncthreads();
        // compaction: wave w owns queries 2w, 2w+1
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int qq = wave * 2 + u;
            const int c = L.cnt[qq];
            const bool last = rd + 1 == rounds;
            if (c > KP || (last && c > 0)) {
                uint64_t best = KEY_NONE;
                for (int base = 0; base < c; base += 64) {
                    const uint64_t key = (base + lane < c) ? L.cand[qq][base + lane] : KEY_NONE;
                    best = (base == 0) ? wave_sort_desc(key) : wave_merge_top64(best, key, lane);
                }
                if (lane < KP) L.cand[qq][lane] = best;
                if (lane == KP - 1 && c >= KP) L.thr[qq] = orderable_f32((uint32_t)(best >> 32));
                if (lane == 0) L.cnt[qq] = min(c, KP);
            }
        }
        __syncthreads();
    }
    // ---- emit [query][workgroup][KP] (sorted; unused slots -inf / -1)
    for (int e = tid; e < 16 * KP; e += 512) {
        const int qq = e / KP, sidx = e % KP;
        if (qq >= p.nq) continue;
        const size_t o = ((size_t)qq * p.n_chunks + blockIdx.x) * KP + sidx;
        if (sidx < L.cnt[qq]) {
            const uint64_t key = L.cand[qq][sidx];
            p.cand_scores[o] = orderable_f32((uint32_t)(key >> 32));
            p.cand_ids[o] = (int)(~(uint32_t)key);
        } else {
            p.cand_scores[o] = -INFINITY;
            p.cand_ids[o] = -1;
        }
    }
}

template <int KP>
static hipError_t launch_t(const SearchArgs& a, hipStream_t s) {
    const int lds = a.dim * 32 + (int)sizeof(StreamLds);
    auto k = search_stream_kernel<KP>;
    static unsigned long long attr = 0;     // bit d: set on device d
    set_max_dynamic_lds((const void*)k, 2560 * 32 + (int)sizeof(StreamLds), attr);
    int rows = (int)((a.n_docs + SS_WG - 1) / SS_WG);
    rows = (rows + 15) / 16 * 16;
    hipLaunchKernelGGL(k, dim3(SS_WG), dim3(512), lds, s, a, rows);
    return hipGetLastError();
}

hipError_t launch_search_stream(const SearchArgs& a, int kp, hipStream_t s) {
    if (a.n_chunks != SS_WG) return hipErrorInvalidValue;
    switch (kp) {
        case 16: return launch_t<16>(a, s);
        case 32: return launch_t<32>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vr
