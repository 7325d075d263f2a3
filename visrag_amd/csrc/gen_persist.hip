// The EVisRAG generator's decode step, all decoder layers in ONE persistent launch (one workgroup per CU).
//
// As separate launches (gen.hip: gen_layer, decode) a layer is 4 weight-streaming GEMMs (gemm_skinny.hip) and 4 small
// kernels; every weight stream already moves at 680-770 GB/s per XCD once it is up, what the step loses is the fixed cost of
// 224 launches: request ramp and drain of each GEMM (~4 us) and ~5 us of pure latency per small kernel — about a third of
// the 3.29 ms token.  Here the same arithmetic, in the same order (the results are the launches' results bit for bit:
// tests/test_gpu_decode_persist.py), runs as PHASES of one kernel separated by grid barriers, and the weights do not wait
// for the activations: before a workgroup waits at a barrier it has already requested the first three K-steps of the NEXT
// projection's weights into its LDS ring, so the HBM stays busy under the barrier and the small phases.
//
// MEASURED (7B shape, 1405-row cache; per layer, workgroup 0's clock): the four streams 65 us — gate|up 272 MB in 42 us
// and down 136 MB in 17 us + its prefetch: 6.5 TB/s, against 5.4 / 5.7 as launches —, the small phases 16 us (R 1.2 + 2.0,
// A rows 1.0 + 2.1 + 0.6, range merge 2.2, attention 6.9 incl. q / k / v + mRoPE), and SEVEN BARRIERS 30 us: 1.6-2.2 us
// each when the memory system is quiet (two-level arrival counters, tools/probe_gridbar.hip: 2.3), 3.8-4.9 us under the
// 24 MB prefetch burst they are there to overlap (arrival atomic, forward atomic and poll are three loaded memory round
// trips).  111 us per layer = 3.33 ms per token against 3.29 ms as launches: level, so the kernel is OPT-IN
// (VR_DECODE_PERSIST=1).  What would make it win is fewer dependent round trips per hand-over (data-as-flag hand-overs
// instead of store-acknowledge -> atomic -> poll), not more bytes in flight: pulling two to four more K-steps into the L2
// with touch-ahead loads during the barriers measured 3.40-3.42 ms.
//
//   layer l:   [R: h += sum of the down planes of l-1; row sum of squares]  B  [q|k|v: A row = rmsnorm(h) built by every
//              workgroup for its K range; planes]  B  [per (KV head, KV range): q / k / v from the planes + bias, mRoPE, cache
//              append, flash attention over the range -> partial rows + log-sum-exps]  B  [o: A row = the ranges merged;
//              planes]  B  [R]  B  [gate|up: A = rmsnorm(h); SwiGLU in the epilogue -> act]  B  [down: planes]  B
//   end:       [R; the next kernel's A row = rmsnorm(h) with the final norm's weight]
//
// Memory model.  The workgroups sit on eight XCDs with one L2 each; inside a launch nothing keeps those coherent, and a
// device-scope fence (buffer_wbl2 + buffer_inv) costs tens of microseconds here.  So every word one
// workgroup writes for another to read in the SAME launch — h, the sums of squares, partial planes, act, partial attention
// rows — is stored and loaded as a relaxed device-scope atomic (sc1: written through / read past the L2's non-coherent lines;
// probe_gridbar.hip exchanges 65 k words per barrier that way without one stale read), ordered by `s_waitcnt vmcnt(0)`
// before the barrier's arrival.  Weights, norm weights, biases and GenState are launch-constant: plain loads.  The KV cache
// row of the step is written and read by ONE workgroup (the last range of its KV head): same CU, same L2.
// Roofline: HBM (the layers' weights, 13.1 GB for the 7B model, once per token).
#include "attention_body.h"
#include "gemm_core.h"
#include "gen_math.h"
#include "kernels.h"

namespace vr {

namespace {

constexpr int PS_STAGE = 256 * 128;                     // W stage: 256 rows x 64 k (gemm_skinny.hip's image and swizzle)
constexpr int PS_RING = 4 * PS_STAGE;
constexpr int PS_ABUF = 8192;                           // the unit's A row: <= 4096 bf16
constexpr int PS_MISC = 256;                            // flags
constexpr int PS_QROWS = 16 * 128 * 2;                  // the attention unit's query rows (group <= 16 heads x 128) bf16
constexpr int PS_SMEM = PS_RING + PS_ABUF + PS_MISC + PS_QROWS;
constexpr unsigned PS_OOB = 0x80000000u;
constexpr int PS_W_AUX = 2;                             // nt, as gemm_skinny.hip
static_assert(PS_STAGE + PS_ABUF >= attn_smem_bytes<128, 0>(), "the attention phase lives in stage 3 + the A row");

__device__ __forceinline__ float ldc(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ldc(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16 bytes at a time with the same cache policy (aux 16 = sc1) through a buffer descriptor: a quarter of the requests
constexpr int PS_SC1 = 16;
__device__ __forceinline__ auto coh_rsrc(const void* base) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000); }
template <typename R> __device__ __forceinline__ f32x4 ldc4(R rs, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, PS_SC1));
}
template <typename R> __device__ __forceinline__ u32x4 ldc4u(R rs, unsigned byte_off) { return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, PS_SC1); }
template <typename R> __device__ __forceinline__ float ldc1(R rs, unsigned byte_off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, PS_SC1));
}
template <typename R> __device__ __forceinline__ void stc4(R rs, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, byte_off, 0, PS_SC1);
}

// sync words (unsigned long long): [0] global arrivals, [8 + 8 g] arrivals of group g (= workgroup % 8: its XCD), [96] launches
// completed (the barrier targets of a launch start from it), [104] abort
constexpr int SY_GLOBAL = 0, SY_GROUP = 8, SY_GEN = 96, SY_ABORT = 104;

struct Barrier {
    unsigned long long* sync;
    unsigned long long base;          // barriers completed before this launch
    unsigned* abort_host;
    int k = 0;                        // barriers of this launch passed so far
    bool dead = false;
    // every wave: its stores acknowledged; lane 0 of wave 3: arrive (the group's last arriver bumps the global counter).
    // Two levels: eight group counters (group = workgroup % 8) take 32 arrivals each instead of one taking 256 (2.3 against
    // 4.0 us on a quiet chip, tools/probe_gridbar.hip; one level with every workgroup polling the eight counters: 3.57 ms
    // per token against 3.33).  Wave 3 also polls, and takes no part in the prefetch between arrive() and wait(): a wave's
    // loads return in order, a poll behind 24 LDS-DMA requests would not see the counter before those have landed.
    __device__ __forceinline__ void arrive() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ++k;
        if (threadIdx.x == 192 && !dead) {
            const unsigned g = blockIdx.x & 7u, gsz = (gridDim.x - g + 7u) / 8u;
            const unsigned long long K = base + (unsigned long long)k;
            const unsigned long long old = __hip_atomic_fetch_add(sync + SY_GROUP + 8 * g, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == K * gsz) __hip_atomic_fetch_add(sync + SY_GLOBAL, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __device__ __forceinline__ void wait(int* flag_lds) {
        if (threadIdx.x == 192) {
            const unsigned long long target = (base + (unsigned long long)k) * min(8u, gridDim.x);
            unsigned spins = 0;
            while (!dead && __hip_atomic_load(sync + SY_GLOBAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 20) || __hip_atomic_load(sync + SY_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(sync + SY_ABORT, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (abort_host) __hip_atomic_store(abort_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    dead = true;
                }
            }
            *flag_lds = dead;
        }
        __syncthreads();
        dead = __builtin_amdgcn_readfirstlane(*flag_lds) != 0;     // (uniform: an LDS load alone would make every branch on it divergent)
        __syncthreads();
    }
};

}  // namespace

__global__ __launch_bounds__(256) void decode_persist_kernel(PersistArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const abuf = smem + PS_RING;
    float* const misc = reinterpret_cast<float*>(smem + PS_RING + PS_ABUF);
    bf16_t* const qrows = reinterpret_cast<bf16_t*>(smem + PS_RING + PS_ABUF + PS_MISC);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int bid = blockIdx.x;
    const GenState* st = p.st;

    Barrier bar;
    bar.sync = p.sync;
    bar.abort_host = p.abort_host;
    bar.base = __hip_atomic_load(p.sync + SY_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * (unsigned long long)(7 * p.n_layers);

    // ---- a streaming unit: 256 output columns x a K range of one weight matrix
    // (every field wave-uniform BY CONSTRUCTION — readfirstlane — so that the descriptor and the scalar offsets of the LDS-DMA
    // loads sit in SGPRs; left to its own analysis hipcc wraps each load in a waterfall loop)
    struct Unit { unsigned wlo, whi, ldw2; int n0, split, nk; unsigned kof; int on; };
    auto rfl = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto unit_of = [&](const void* W, int ldw, int N, int K, int ksplit) {
        Unit u{};
        const int tiles = (N + 255) / 256, ks = ksplit > 1 ? ksplit : 1;
        u.on = rfl(bid < tiles * ks);
        const int b = u.on ? bid : 0;
        u.split = rfl(b / tiles);
        u.n0 = rfl((b - u.split * tiles) * 256);
        const int nk_all = K / GEMM_BK, per = (nk_all + ks - 1) / ks;
        const int k0 = u.split * per;
        u.nk = rfl(max(0, min(per, nk_all - k0)));
        u.kof = (unsigned)rfl(k0 * GEMM_BK);
        u.ldw2 = (unsigned)rfl(ldw * 2);
        const unsigned long long wa = (unsigned long long)((const char*)W + ((size_t)u.n0 * ldw + u.kof) * 2);
        u.wlo = (unsigned)rfl((int)(unsigned)wa);
        u.whi = (unsigned)rfl((int)(unsigned)(wa >> 32));
        return u;
    };
    const unsigned lchunk = (unsigned)(((lane & 7) ^ (lane >> 3)) << 4);
    // the 8 loads of K-step kt of unit u into stage kt % 4 (this wave's 64 rows)
    auto w_issue = [&](const Unit& u, int kt) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((unsigned long long)u.whi << 32 | u.wlo), 0, 0x7FFFFFFF, 0x00020000);
        char* stg = smem + (kt & 3) * PS_STAGE + wave * 8192;
        const unsigned kb = kt < u.nk ? (unsigned)kt * (GEMM_BK * 2) : PS_OOB;
        const unsigned lof = (unsigned)(lane >> 3) * u.ldw2 + lchunk;
        const unsigned rg = u.ldw2 * 8u;
#pragma unroll
        for (int d = 0; d < 8; ++d)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, VR_LDS(stg + d * 1024), 16, lof + kb, (unsigned)rfl((int)(((unsigned)wave * 8u + d) * rg)), 0, PS_W_AUX);
    };
    // the first three K-steps of unit u, requested by waves 0..2 for all four waves' rows (96 instructions, 32 each)
    // the first K-steps of unit u, requested by waves 0..2 for all four waves' rows (32 instructions per K-step): three when the
    // phase in between needs stage 3 of the ring (the attention tiles, the range merge), else all four stages
    auto prefetch_n = [&](const Unit& u, auto steps_c) {
        constexpr int STEPS = decltype(steps_c)::value;
        if (!u.on || wave == 3) return;
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((unsigned long long)u.whi << 32 | u.wlo), 0, 0x7FFFFFFF, 0x00020000);
        const unsigned lof = (unsigned)(lane >> 3) * u.ldw2 + lchunk;
        const unsigned rg = u.ldw2 * 8u;
#pragma unroll
        for (int i = 0; i < (STEPS * 32 + 2) / 3; ++i) {
            const int j = wave + 3 * i, kt = j >> 5, rgi = j & 31;          // K-step, 8-row group of the 256 rows
            if (j >= STEPS * 32) continue;
            const unsigned kb = kt < u.nk ? (unsigned)kt * (GEMM_BK * 2) : PS_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, VR_LDS(smem + kt * PS_STAGE + rgi * 1024), 16, lof + kb, (unsigned)rfl((int)((unsigned)rgi * rg)), 0, PS_W_AUX);
        }
    };
    auto prefetch = [&](const Unit& u) { prefetch_n(u, std::integral_constant<int, 3>{}); };
    auto prefetch4 = [&](const Unit& u) { prefetch_n(u, std::integral_constant<int, 4>{}); };
    // the K loop over a prefetched unit with the A row (row 0 of the MFMA's 16) in abuf: gemm_skinny.hip's MFMA order
    const int ch0 = (fq ^ (fr & 7)) << 4, ch1 = ((4 + fq) ^ (fr & 7)) << 4;
    auto stream = [&](const Unit& u, f32x4 (&acc)[4], bool pre4 = false) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // K-steps 0..2 (0..3: pre4) are in the ring (prefetched before the barrier and waited for here: every wave starts the loop with an
        // empty request queue, so its vmcnt(16) keeps meaning "my loads of step kt have landed")
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int kt = 0; kt < u.nk; ++kt) {
            VR_WAIT_VM_BARRIER(16);                 // K-step kt has landed everywhere; everyone is done with stage (kt - 1) % 4
            if (!(pre4 && kt == 0)) w_issue(u, kt + 3);   // (four steps prefetched: step 3 is there.  Past the end: zeros into a dead stage, no traffic — the count above stays valid)
            const char* wr = smem + (kt & 3) * PS_STAGE + (wave * 64 + fr) * 128;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(abuf + kt * 128 + fq * 16);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(abuf + kt * 128 + 64 + fq * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wr + j * 2048 + ch0);
                const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wr + j * 2048 + ch1);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, a0, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, a1, acc[j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the steps requested past the end: zeros into dead stages)
        __syncthreads();                                    // the ring is free for the next unit's prefetch
    };
    // fp32 plane of a split: out[split][n] (M = 1)
    auto store_plane = [&](const Unit& u, const f32x4 (&acc)[4], float* planes, int N) {
        if (fr != 0) return;
        const auto prs = coh_rsrc(planes);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = u.n0 + wave * 64 + j * 16 + fq * 4;
            if (n < N) stc4(prs, ((unsigned)u.split * (unsigned)N + (unsigned)n) * 4u, acc[j]);
        }
    };
    // R: h += planes (ordered sum, rmsnorm_accum_row_kernel's), sum of squares of wave w's 64 float4 columns -> ss[w]
    const int nv = p.E >> 2;
    f32x4 rv = {0.f, 0.f, 0.f, 0.f};                       // this lane's column of h after the last R (the final norm reuses it)
    auto reduce_rows = [&](const float* planes, int nsplit) {
        if (bid * 64 >= nv || wave != 0) return;
        const int c = bid * 64 + lane;
        const auto prs = coh_rsrc(planes), hrs = coh_rsrc(p.h);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            const unsigned co = (unsigned)c * 16u, ps = (unsigned)p.E * 4u;
            const f32x4 h0 = ldc4(hrs, co);
            f32x4 acc = ldc4(prs, co);
            // every plane's load is independent, the sum is ordered: up to 24 in flight, then added in plane order
            int sp = 1;
            for (; sp < nsplit; sp += 24) {
                f32x4 t[24];
#pragma unroll
                for (int i = 0; i < 24; ++i) t[i] = ldc4(prs, co + (unsigned)min(sp + i, nsplit - 1) * ps);
#pragma unroll
                for (int i = 0; i < 24; ++i)
                    if (sp + i < nsplit) acc += t[i];
            }
            v = h0 + 1.0f * acc;
            stc4(hrs, co, v);
        }
        rv = v;
        const float s = wave_sum(sumsq4(v));
        if (lane == 0) stc(p.ss + bid, s);
    };
    // rstd of the row from the 16 partial sums, added in index order; every wave for itself (no LDS, no barrier)
    auto row_rstd = [&]() {
        const float mine = ldc(p.ss + (lane & 15));
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += __shfl(mine, i, 64);
        return rsqrtf(tot / p.E + p.eps);
    };
    // abuf[0 .. n) = bf16(rmsnorm(h)[k0 .. k0 + n) * g): the launches' expression (v * rstd) * w; n <= 4096
    auto norm_to_abuf = [&](const float* g, unsigned k0, int n) {
        const auto hrs = coh_rsrc(p.h);
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ldc4(hrs, (k0 + (unsigned)min((tid + i * 256) * 4, max(n - 4, 0))) * 4u);
        const float rstd = row_rstd();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + i * 256;
            if (c * 4 < n) {
                const f32x4 y = v[i] * rstd * *reinterpret_cast<const f32x4*>(g + k0 + (size_t)c * 4);
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(y[r]);
                *reinterpret_cast<bf16x4*>(abuf + (size_t)c * 8) = o;
            }
        }
        __syncthreads();
    };

    int* const flag = reinterpret_cast<int*>(misc + 8);
    Unit next = unit_of(p.layers[0].wqkv, p.ldw_qkv, p.QKV, p.E, p.ks_qkv);
    prefetch4(next);
    for (int l = 0; l < p.n_layers && !bar.dead; ++l) {
        // (pointers from the table made wave-uniform by hand: descriptors built from them must sit in SGPRs)
        auto uptr = [&](const void* q) {
            const unsigned long long a = (unsigned long long)q;
            return (void*)((unsigned long long)(unsigned)rfl((int)(unsigned)(a >> 32)) << 32 | (unsigned)rfl((int)(unsigned)a));
        };
        const PersistLayer Lr = p.layers[l];
        const PersistLayer L = {uptr(Lr.wqkv), uptr(Lr.wo), uptr(Lr.wgu), uptr(Lr.wd), (const float*)uptr(Lr.bqkv), (const float*)uptr(Lr.g1),
                                (const float*)uptr(Lr.g2), uptr(Lr.kc), uptr(Lr.vc)};
        f32x4 acc[4];
        // ---------------- R (layers > 0: the previous layer's down planes)
        if (l > 0) {
            reduce_rows(p.planes, p.ks_d);
            bar.arrive(); bar.wait(flag);
        }
        // ---------------- q | k | v
        {
            const Unit u = next;
            if (u.on) {
                if (l == 0) {                              // the step's first A row was made by the launch before this one
                    for (int c = tid; c * 8 < u.nk * 64; c += 256)
                        *reinterpret_cast<u32x4*>(abuf + (size_t)c * 16) = *reinterpret_cast<const u32x4*>((const char*)p.xn + (u.kof + (size_t)c * 8) * 2);
                    __syncthreads();
                } else {
                    norm_to_abuf(L.g1, u.kof, u.nk * 64);
                }
                stream(u, acc, true);
                store_plane(u, acc, p.planes, p.QKV);
            }
            next = unit_of(L.wo, p.ldw_o, p.E, p.QD, p.ks_o);
            bar.arrive();
            prefetch(next);                                // stages 0..2: the attention phase below lives in stage 3 + the A row
            bar.wait(flag);
        }
        // ---------------- attention: unit = (KV range b, KV head hk); q rows of the group built here, k / v of the new
        //                  row appended by the unit of the LAST range
        {
            const int G = p.H / p.KV;
            const int splits = st->splits;
            const int b = bid / p.KV, hk = bid - b * p.KV;
            if (bid < GEN_ATT_SPLITS * p.KV && b < splits) {
                const bool last = b == splits - 1;
                const int nslot = G + (last ? 2 : 0);      // G query heads (+ the key and the value head)
                // mrope_cache_kernel's arithmetic per head slot (q heads hk * G + sl, then the k head, then the v head); a wave
                // takes slots wave, wave + 4, ...: three at a time, their plane loads (two columns per lane) all in flight together
                const auto prs = coh_rsrc(p.planes);
                for (int s0 = wave; s0 < nslot; s0 += 12) {
                    float x1[3], x2[3];
                    unsigned off[3];
                    int hs[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int sl = min(s0 + 4 * q, nslot - 1);
                        hs[q] = sl < G ? hk * G + sl : (sl == G ? p.H + hk : p.H + p.KV + hk);
                        const int col = hs[q] * 128;
                        x1[q] = L.bqkv ? L.bqkv[col + lane] : 0.f;
                        x2[q] = L.bqkv ? L.bqkv[col + 64 + lane] : 0.f;
                        off[q] = (unsigned)(col + lane) * 4u;
                    }
                    const unsigned ps = (unsigned)p.QKV * 4u;
                    for (int sp = 0; sp < p.ks_qkv; sp += 16) {
                        float a[3][16], c2[3][16];
#pragma unroll
                        for (int q = 0; q < 3; ++q)
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                const unsigned po = (unsigned)min(sp + i, p.ks_qkv - 1) * ps;
                                a[q][i] = ldc1(prs, off[q] + po);
                                c2[q][i] = ldc1(prs, off[q] + po + 256u);
                            }
#pragma unroll
                        for (int q = 0; q < 3; ++q)
#pragma unroll
                            for (int i = 0; i < 16; ++i)
                                if (sp + i < p.ks_qkv) { x1[q] += a[q][i]; x2[q] += c2[q][i]; }
                    }
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int sl = s0 + 4 * q;
                        if (sl >= nslot) continue;
                        float y1 = x1[q], y2 = x2[q];
                        if (hs[q] < p.H + p.KV) {
                            const int c = lane < p.sec_t ? 0 : (lane < p.sec_t + p.sec_h ? 1 : 2);
                            const float pos = (float)st->pos[c];
                            const float ang = pos * p.inv_freq[lane];
                            rope_rotate(y1, y2, cosf(ang), sinf(ang));
                        }
                        bf16_t* dst;
                        if (sl < G) dst = qrows + sl * 128;
                        else if (sl == G) dst = (bf16_t*)L.kc + (size_t)st->len * p.KVD + hk * 128;
                        else dst = (bf16_t*)L.vc + (size_t)st->len * p.KVD + hk * 128;
                        dst[lane] = f2bf(y1);
                        dst[64 + lane] = f2bf(y2);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the cache row is in this XCD's L2 before the tile loads ask for it
                __syncthreads();
                AttnArgs a{};
                // (a.q is never read — q_lds below — but a null there sends hipcc's SimplifyCFG into a crash)
                a.q = qrows; a.ldq = 128; a.k = L.kc; a.ldk = p.KVD; a.v = L.vc; a.ldv = p.KVD;
                a.heads = p.KV; a.head_dim = 128; a.scale = 1.0f / sqrtf(128.0f); a.kv_group = 1; a.q_head_stride = G * 128;
                a.cu_q = st->cu_q; a.cu_kv = st->cu_kv; a.out = p.attp; a.ldo = p.KVD; a.B = GEN_ATT_SPLITS; a.max_q = G;
                a.causal = 0; a.q_shared = 1; a.lse = p.lse;
                attention_body<128, 1, 0, true>(a, b * p.KV + hk, smem + 3 * PS_STAGE, qrows);
            }
            bar.arrive(); bar.wait(flag);
        }
        // ---------------- o projection: A row = the ranges merged by their log-sum-exps (gemm_skinny.hip's COMBINE prologue)
        {
            const Unit u = next;
            if (u.on) {
                // the ranges' partial rows and log-sum-exps of this K range's columns: fetched ONCE per workgroup with 16-byte
                // sc1 loads into stage 3 of the ring (free until the K loop's first request), then one thread per column
                const int cS = st->splits, G = p.H / p.KV;
                const int ncol = u.nk * 64, nch = ncol >> 3;               // columns, 16-byte chunks per range row
                char* tile = smem + 3 * PS_STAGE;                          // bf16 [16][ncol], then f32 lse [16][8 heads]
                float* lse_t = reinterpret_cast<float*>(tile + GEN_ATT_SPLITS * 1024);
                const int h_first = (int)u.kof >> 7;
                {
                    const auto ars = coh_rsrc(p.attp);
                    for (int c = tid; c < GEN_ATT_SPLITS * nch; c += 256) {
                        const int t = c / nch, ch = c - t * nch;
                        const int col = (int)u.kof + ch * 8, h = col >> 7, d = col & 127;
                        const int r = (min(t, max(cS - 1, 0)) * G + h % G) * p.KV + h / G;
                        *reinterpret_cast<u32x4*>(tile + t * 1024 + ch * 16) = ldc4u(ars, ((unsigned)r * 128u + (unsigned)d) * 2u);
                    }
                    if (tid < GEN_ATT_SPLITS * 8) {
                        const int t = tid >> 3, h = min(h_first + (tid & 7), p.H - 1);
                        const int r = (min(t, max(cS - 1, 0)) * G + h % G) * p.KV + h / G;
                        lse_t[tid] = t < cS ? ldc(p.lse + r) : -INFINITY;
                    }
                }
                __syncthreads();
                for (int cc = tid; cc < ncol; cc += 256) {
                    const int hl = (((int)u.kof + cc) >> 7) - h_first;
                    float cl[GEN_ATT_SPLITS], cpv[GEN_ATT_SPLITS];
#pragma unroll
                    for (int t = 0; t < GEN_ATT_SPLITS; ++t) {
                        cl[t] = lse_t[t * 8 + hl];
                        cpv[t] = bf2f(reinterpret_cast<const bf16_t*>(tile + t * 1024)[cc]);
                    }
                    float mx = -INFINITY;
#pragma unroll
                    for (int t = 0; t < GEN_ATT_SPLITS; ++t) mx = fmaxf(mx, cl[t]);
                    float num = 0.f, den = 0.f;
#pragma unroll
                    for (int t = 0; t < GEN_ATT_SPLITS; ++t) {
                        const float e = exp2f(cl[t] - mx);
                        merge_range(num, den, e, cpv[t]);
                    }
                    reinterpret_cast<bf16_t*>(abuf)[cc] = f2bf(num / den);
                }
                __syncthreads();
                stream(u, acc);
                store_plane(u, acc, p.planes, p.E);
            }
            next = unit_of(L.wgu, p.ldw_gu, p.N2, p.E, 1);
            bar.arrive();
            prefetch4(next);
            bar.wait(flag);
        }
        // ---------------- R (o planes), then gate | up with SwiGLU in the epilogue
        reduce_rows(p.planes, p.ks_o);
        bar.arrive(); bar.wait(flag);
        {
            const Unit u = next;
            if (u.on) {
                norm_to_abuf(L.g2, 0, p.E);
                stream(u, acc, true);
                if (fr == 0) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int n = u.n0 + wave * 64 + jj * 32;          // a [16 gate | 16 up] block pair = 16 columns of act
                        if (n < p.N2) {
                            const f32x4 g = acc[2 * jj], uu = acc[2 * jj + 1];
                            bf16x4 o;
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = f2bf(g[r] / (1.0f + __expf(-g[r])) * uu[r]);   // (swiglu_sum_kernel's expression)
                            const u32x2 w2 = __builtin_bit_cast(u32x2, o);
                            unsigned* dst = reinterpret_cast<unsigned*>((bf16_t*)p.act + n / 2 + fq * 4);
                            stc(dst, w2[0]); stc(dst + 1, w2[1]);
                        }
                    }
                }
            }
            next = unit_of(L.wd, p.ldw_d, p.E, p.Ip, p.ks_d);
            bar.arrive();
            prefetch4(next);
            bar.wait(flag);
        }
        // ---------------- down projection
        {
            const Unit u = next;
            if (u.on) {
                for (int c = tid; c * 2 < u.nk * 64; c += 256)
                    reinterpret_cast<unsigned*>(abuf)[c] = ldc(reinterpret_cast<const unsigned*>(p.act) + u.kof / 2 + c);
                __syncthreads();
                stream(u, acc, true);
                store_plane(u, acc, p.planes, p.E);
            }
            const bool more = l + 1 < p.n_layers;
            if (more) next = unit_of(p.layers[l + 1].wqkv, p.ldw_qkv, p.QKV, p.E, p.ks_qkv);
            bar.arrive();
            if (more) prefetch4(next);
            bar.wait(flag);
        }
    }
    // ---------------- the last R and the A row of the lm_head: rmsnorm(h) * final norm weight (the columns this wave reduced)
    if (!bar.dead) {
        reduce_rows(p.planes, p.ks_d);
        bar.arrive(); bar.wait(flag);
        if (bid * 64 < nv && wave == 0 && !bar.dead) {
            float tot = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) tot += ldc(p.ss + i);
            const float rstd = rsqrtf(tot / p.E + p.eps);
            const int c = bid * 64 + lane;
            if (c < nv) {
                const f32x4 y = rv * rstd * reinterpret_cast<const f32x4*>(p.g_final)[c];
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = f2bf(y[r]);
                reinterpret_cast<bf16x4*>(p.xn)[c] = o;
            }
        }
    }
    // the launch is over for this workgroup once it has passed the last barrier; workgroup 0 counts the launch
    if (bid == 0 && tid == 0) __hip_atomic_fetch_add(p.sync + SY_GEN, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

hipError_t launch_decode_persist(const PersistArgs& a, int grid, hipStream_t s) {
    static unsigned long long attr = 0;
    set_max_dynamic_lds((const void*)decode_persist_kernel, PS_SMEM, attr);
    hipLaunchKernelGGL(decode_persist_kernel, dim3(grid), dim3(256), PS_SMEM, s, a);
    return hipGetLastError();
}

// workgroups of the kernel that fit one CU (0: the launch must not be attempted)
int decode_persist_occupancy() {
    static unsigned long long attr = 0;
    set_max_dynamic_lds((const void*)decode_persist_kernel, PS_SMEM, attr);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, decode_persist_kernel, 256, PS_SMEM) != hipSuccess) return 0;
    return n;
}

}  // namespace vr
