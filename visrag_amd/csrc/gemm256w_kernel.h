// The kernel text of gemm256w.hip (W_KERNEL_TEMPLATE / W_KERNEL_NAME are set by the includer).
W_KERNEL_TEMPLATE
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void W_KERNEL_NAME(GemmArgs p) {
    static_assert(NJ == 8 || (NJ == 6 && !PLAIN), "wave tile 128 x 128 or 128 x 96");
    constexpr int BN = 32 * NJ;                 // tile columns
    constexpr int NS = 8 * NJ;                  // MFMAs per phase (slots)
    constexpr int NR = 8 + NJ;                  // fragment reads per k-half = LDS-DMA loads per K-step and wave
    constexpr int DS = NJ == 8 ? 5 : 4;         // one load per DS slots
    constexpr int SB1 = NS * 5 / 8;             // slot of the phase-1 barrier (the k-half-1 reads end at slot 2 NR - 2)
    constexpr int D1 = (NS - 1 - (SB1 + 2)) / DS + 1;   // loads issued in phase 1 (slots SB1 + 2, + DS, ...)
    constexpr int SB2 = NJ;                     // slot of the phase-2 barrier
    constexpr int NF = NJ / 2;                  // fragments per epilogue piece: 64 or 48 columns
    static_assert(SB2 + 3 + DS * (NR - D1 - 1) < NS && SB2 + 2 + 2 * (NR - 1) < NS && 2 * NR - 2 < SB1, "schedule fits");
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef VR_W_TIMING                   // tile anatomy (tools/w_anatomy.py, tagged builds only, one tile per workgroup)
    const unsigned long long tm0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long tm1 = 0, tm2 = 0, cy1 = 0, cy2 = 0;
#endif
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + G256_BM - 1) / G256_BM;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int total = tiles_m * tiles_n * ks;
    const int Ks = p.K / ks;
    const int nk = Ks / GEMM_BK;

    const int my_first = xcd_remap(blockIdx.x, total);
    // tile index -> (m0, n0, split); grouped rasterisation as in gemm256_bf16_kernel
    const int GM = p.raster_gm > 0 ? p.raster_gm : 1;
    auto coords = [&](int tt, int& m0, int& n0, int& split) {
        split = tt / (tiles_m * tiles_n);
        const int t = tt - split * (tiles_m * tiles_n);
        const int gsz = GM * tiles_n;
        const int g = t / gsz, r = t % gsz;
        const int gm = min(GM, tiles_m - g * GM);
        m0 = __builtin_amdgcn_readfirstlane((g * GM + r % gm) * G256_BM);
        n0 = __builtin_amdgcn_readfirstlane((r / gm) * BN);
        split = __builtin_amdgcn_readfirstlane(split);
    };

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;

    // ---- LDS-DMA addressing: wave w fills rows [64 w, 64 w + 64) of the A tile and [8 NJ w, 8 NJ (w + 1)) of
    //      the W tile, 8 rows per instruction; lane l -> row l / 8, 16-byte chunk (l % 8) ^ (row % 8) of the
    //      128-byte k-slice.  One descriptor per MATRIX (both below 2 GiB, checked by the launcher); a tile's
    //      byte offset and the K offset go into the per-lane offset, the row group into the scalar offset.
    const auto arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7FFFFFFF, 0x00020000);
    const auto wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7FFFFFFF, 0x00020000);
    const unsigned lchunk = (unsigned)(((lane & 7) ^ (lane >> 3)) << 4);
    const unsigned lofA = (unsigned)(lane >> 3) * (unsigned)p.lda * 2u + lchunk;
    const unsigned lofW = (unsigned)(lane >> 3) * (unsigned)p.ldw * 2u + lchunk;
    const unsigned rgA = (unsigned)p.lda * 16u, rgW = (unsigned)p.ldw * 16u;      // bytes per 8-row group
    const unsigned sA0 = (unsigned)wave * 8u * rgA, sW0 = (unsigned)wave * (unsigned)NJ * rgW;
    char* const dmaA = smem + wave * 8192;
    char* const dmaW = smem + G256_TILE_BYTES + wave * (NJ * 1024);
    auto tile_off_a = [&](int m0, int split) { return ((unsigned)m0 * (unsigned)p.lda + (unsigned)split * (unsigned)Ks) * 2u; };
    auto tile_off_w = [&](int n0, int split) { return ((unsigned)n0 * (unsigned)p.ldw + (unsigned)split * (unsigned)Ks) * 2u; };

    // one of the NR loads of a K-step: d < 8 -> A row group d, else W row group d - 8
    auto dma = [&](int stage, int d, unsigned vA, unsigned vW) {
        if (d < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, VR_LDS(dmaA + stage * W_STAGE + d * 1024), 16, vA, sA0 + d * rgA, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, VR_LDS(dmaW + stage * W_STAGE + (d - 8) * 1024), 16, vW, sW0 + (d - 8) * rgW, 0, 0);
    };

    // ---- fragment addressing: row = strip * 16 + fr, chunk (kk * 4 + fq) ^ (row & 7).  One LDS pointer per
    //      (operand, stage, k-half) held in a VGPR; the strip is an immediate offset (strip * 2 KiB).
    typedef const __attribute__((address_space(3))) bf16x8* frag_p;
    frag_p pA[2][2], pW[2][2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ch = ((kk * 4 + fq) ^ (fr & 7)) << 4;
            pA[st][kk] = (frag_p)VR_LDS(smem + st * W_STAGE + (wm * 128 + fr) * 128 + ch);
            pW[st][kk] = (frag_p)VR_LDS(smem + st * W_STAGE + G256_TILE_BYTES + (wn * 16 * NJ + fr) * 128 + ch);
            asm volatile("" : "+v"(pA[st][kk]), "+v"(pW[st][kk]));
        }

    bf16x8 a0[8], w0[NJ], a1[8], w1[NJ];
    int m0, n0, split;
    coords(my_first, m0, n0, split);
    unsigned curA = tile_off_a(m0, split), curW = tile_off_w(n0, split);

    // ---- prologue: K-steps 0 and 1 in flight, the accumulators zeroed under their latency (256 register writes:
    //      half a microsecond), k-half-0 fragments of step 0 requested
    {
        const unsigned k1 = nk > 1 ? (unsigned)(GEMM_BK * 2) : W_OOB;
#pragma unroll
        for (int d = 0; d < NR; ++d) dma(0, d, lofA + curA, lofW + curW);
#pragma unroll
        for (int d = 0; d < NR; ++d) dma(1, d, lofA + curA + k1, lofW + curW + k1);
        W_FOR_EACH_ACC(W_ZERO)
        if constexpr (NJ == 8) VR_WAIT_VM_BARRIER(16); else VR_WAIT_VM_BARRIER(14);
#pragma unroll
        for (int j = 0; j < NJ; ++j) w0[j] = pW[0][0][j * 128];
#pragma unroll
        for (int i = 0; i < 8; ++i) a0[i] = pA[0][0][i * 128];
    }
#ifdef VR_W_TIMING
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tm1 = __builtin_amdgcn_s_memrealtime();
    cy1 = __builtin_amdgcn_s_memtime();
#endif

    {
        // the loads of the two K-steps past the end of K go nowhere (out of the descriptors' range)
        const unsigned nxtA = W_OOB, nxtW = W_OOB;

        auto step = [&](auto stage_c, int kt) {
            constexpr int S = decltype(stage_c)::value;
            const int k2 = kt + 2;
            const unsigned vA = lofA + (k2 < nk ? curA + (unsigned)k2 * (GEMM_BK * 2) : nxtA + (unsigned)(k2 - nk) * (GEMM_BK * 2));
            const unsigned vW = lofW + (k2 < nk ? curW + (unsigned)k2 * (GEMM_BK * 2) : nxtW + (unsigned)(k2 - nk) * (GEMM_BK * 2));
            __builtin_amdgcn_sched_barrier(0);
            // one auxiliary operation may follow each MFMA; sl = its slot in the phase
            auto aux1 = [&](int sl) {
                if (sl < 2 * NR && (sl & 1) == 0) {
                    const int q = sl >> 1;
                    if (q < NJ) w1[q] = pW[S][1][q * 128];
                    else a1[q - NJ] = pA[S][1][(q - NJ) * 128];
                }
                if (sl == SB1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (sl >= SB1 + 2 && (sl - (SB1 + 2)) % DS == 0) dma(S, (sl - (SB1 + 2)) / DS, vA, vW);     // loads 0 .. D1-1
                __builtin_amdgcn_sched_barrier(0);
            };
            auto aux2 = [&](int sl) {
                if (sl == SB2) { if constexpr (D1 == 5) VR_WAIT_VM_BARRIER(5); else VR_WAIT_VM_BARRIER(4); }
                if (sl >= SB2 + 3 && (sl - (SB2 + 3)) % DS == 0 && D1 + (sl - (SB2 + 3)) / DS < NR)
                    dma(S, D1 + (sl - (SB2 + 3)) / DS, vA, vW);                                              // loads D1 .. NR-1
                if (sl >= SB2 + 2 && sl < SB2 + 2 + 2 * NR && ((sl - SB2) & 1) == 0) {
                    const int q = (sl - (SB2 + 2)) >> 1;
                    if (q < NJ) w0[q] = pW[S ^ 1][0][q * 128];
                    else a0[q - NJ] = pA[S ^ 1][0][(q - NJ) * 128];
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            // ---- phase 1: k-half 0, phase 2: k-half 1 (accumulator n: strip n / 8, fragment n % 8 < NJ)
#define W_P1(n, R, C0, C1, C2, C3) \
            if (((n) & 7) < NJ) { W_MFMA(R, C0, C1, C2, C3, w0[(n) & 7], a0[(n) >> 3]); aux1(((n) >> 3) * NJ + ((n) & 7)); }
#define W_P2(n, R, C0, C1, C2, C3) \
            if (((n) & 7) < NJ) { W_MFMA(R, C0, C1, C2, C3, w1[(n) & 7], a1[(n) >> 3]); aux2(((n) >> 3) * NJ + ((n) & 7)); }
            W_FOR_EACH_ACC(W_P1)
            W_FOR_EACH_ACC(W_P2)
#undef W_P1
#undef W_P2
        };

        for (int kt = 0; kt < nk; kt += 2) {
            step(std::integral_constant<int, 0>{}, kt);
            if (kt + 1 >= nk) break;
            step(std::integral_constant<int, 1>{}, kt + 1);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> VALU reads (see the W_MFMA note)
#ifdef VR_W_TIMING
        tm2 = __builtin_amdgcn_s_memrealtime();
        cy2 = __builtin_amdgcn_s_memtime();
#endif

        // ---- epilogue of tile (m0, n0, split).  The stages may already be receiving the next tile: staging goes
        //      through this wave's 4 KiB above them, one piece at a time (a wave's LDS operations execute in order).
        //      Pieces of 32 accumulator registers are read back into VGPRs one after the other; the scheduling
        //      barriers keep hipcc from pulling later read-backs up and running out of VGPRs.
        const int mrow0 = m0 + wm * 128, nb0 = n0 + wn * (16 * NJ);
        char* const wl0 = smem + 2 * W_STAGE + wave * W_STAGING;
        bool plain_resid = false;
        if constexpr (EPI == EPI_RESID) plain_resid = !p.rowmap;
        // fp32 tiles without a column edge (every encode-path shape) leave through BUFFER loads / stores: the descriptor
        // starts at the tile's first row and ends with its last valid one (rows >= M are out of range: loads return 0,
        // stores are dropped), a lane's offset is one VGPR per 16-row group (the hardware's range check covers the VGPR and
        // immediate offsets, NOT the scalar one — the row step must not go there), the fragment column an immediate.
        // No predicate, no clamp, no 64-bit address arithmetic, no branch: with the per-store `if (m < M && n < N)` of the
        // general form below hipcc puts `s_waitcnt vmcnt(0)` in front of every piece (it counts loads exactly only in
        // straight-line code) and every store is a `flat_store` behind an exec-mask branch.  Round 5, tools/w_anatomy.py
        // and tools/probe_store.hip: residual epilogue 12.2 -> 10.9 us per proj / fc2 tile, fp32 epilogue -> 7.5 us.
        // What is left is the memory system, not this code: 16 rows x 64 B per instruction (the MFMA C layout) moves
        // 33 GB/s per CU in stores and 20 GB/s in read-modify-write even with ONE CU running, rows of >= 256 B move
        // 118 / 33 — but with all 256 CUs in their epilogues together (equal tiles, one round = one phase) the chip
        // gives 27 GB/s per CU in stores and 13 in read-modify-write whatever the shape (7 TB/s of HBM writes; the
        // residual GEMMs do better than that only because part of the stream is still in the memory-side cache).
        // (descriptors must sit in scalar registers: whatever hipcc keeps of `p` in private memory comes back in VGPRs,
        // and a buffer access with a VGPR descriptor is wrapped in a waterfall loop)
        auto uni32 = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
        auto uni_ptr = [&](const void* q) {
            const unsigned long long a = (unsigned long long)q;
            return (void*)(((unsigned long long)uni32((unsigned)(a >> 32)) << 32) | uni32((unsigned)a));
        };
        bool buf_tile = false;
        if constexpr (EPI == EPI_RESID) buf_tile = plain_resid && n0 + BN <= p.N;
        if constexpr (EPI == EPI_F32) buf_tile = !p.rowmap && !p.rowbias && n0 + BN <= p.N;
        if constexpr (EPI == EPI_RESID || EPI == EPI_F32) {
            if (buf_tile) {
                constexpr int MI = 2;           // pieces of 32 rows x 16 NF columns: 8 per wave
                const float alpha = __builtin_bit_cast(float, uni32(__builtin_bit_cast(unsigned, p.alpha)));
                const unsigned ldb = uni32((unsigned)p.ldo * 4u);
                const unsigned nrec = uni32((unsigned)min(G256_BM, p.M - m0) * ldb);
                float* obase = (float*)p.out + (size_t)m0 * p.ldo;
                if constexpr (EPI == EPI_F32) obase += (size_t)split * p.split_stride;
                const auto ors = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(obase), 0, nrec, 0x00020000);
                const bool with_bias = p.bias && (EPI == EPI_RESID || split == 0);
                const auto brs = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(p.bias), 0, uni32(with_bias ? (unsigned)p.N * 4u : 0u), 0x00020000);
                const unsigned voff = (unsigned)(wm * 128 + fr) * ldb + (unsigned)(nb0 + fq * 4) * 4u;
                f32x4 bias[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    bias[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (unsigned)(nb0 + fq * 4) * 4u + j * 64, 0, 0));
                if constexpr (EPI == EPI_RESID) {
                    // out = resid + alpha * (acc + bias), in place; the residual of piece q + RD is in flight while piece q
                    // is combined and stored
                    constexpr int RD = NJ == 6 ? 3 : 2;
                    const auto rrs = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(p.resid + (size_t)m0 * p.ldo), 0, nrec, 0x00020000);
                    f32x4 rs[RD + 1][MI][NF];
                    auto load_piece = [&](int q, f32x4 (&dst)[MI][NF]) {
                        const int h = q & 1, sg = q >> 1;
#pragma unroll
                        for (int i = 0; i < MI; ++i)
#pragma unroll
                            for (int j = 0; j < NF; ++j)
                                dst[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                    rrs, voff + (unsigned)((sg * MI + i) * 16) * ldb + (h * NF + j) * 64, 0, 0));
                    };
#pragma unroll
                    for (int q = 0; q < RD; ++q) load_piece(q, rs[q]);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int h = q & 1, sg = q >> 1;
                        if (q + RD < 8) load_piece(q + RD, rs[(q + RD) % (RD + 1)]);
                        f32x4 acc[MI][NF];
#define W_RD(n, R, C0, C1, C2, C3) \
                        if (((n) & 7) < NJ && ((n) & 7) / NF == h && ((n) >> 4) == sg) W_READ(acc[((n) >> 3) & 1][((n) & 7) % NF], C0, C1, C2, C3);
                        W_FOR_EACH_ACC(W_RD)
#undef W_RD
#pragma unroll
                        for (int i = 0; i < MI; ++i)
#pragma unroll
                            for (int j = 0; j < NF; ++j) {
                                const f32x4 v = rs[q % (RD + 1)][i][j] + alpha * (acc[i][j] + bias[h * NF + j]);
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ors,
                                                                       voff + (unsigned)((sg * MI + i) * 16) * ldb + (h * NF + j) * 64, 0, 0);
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    // fp32 outputs, incl. the split-K partial products (split s writes its own plane, the bias goes with split 0)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int h = q & 1, sg = q >> 1;
                        f32x4 acc[MI][NF];
#define W_RD(n, R, C0, C1, C2, C3) \
                        if (((n) & 7) < NJ && ((n) & 7) / NF == h && ((n) >> 4) == sg) W_READ(acc[((n) >> 3) & 1][((n) & 7) % NF], C0, C1, C2, C3);
                        W_FOR_EACH_ACC(W_RD)
#undef W_RD
#pragma unroll
                        for (int i = 0; i < MI; ++i)
#pragma unroll
                            for (int j = 0; j < NF; ++j)
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j] + bias[h * NF + j]), ors,
                                                                       voff + (unsigned)((sg * MI + i) * 16) * ldb + (h * NF + j) * 64, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        if constexpr (EPI == EPI_RESID) {
            // the general form (a column edge): out = resid + alpha * (acc + bias), fp32, in place, piece by piece.  Addresses
            // are clamped for rows >= M / columns >= N, the stores predicated.
            if (plain_resid && !buf_tile) {
                constexpr int MI = 2;           // pieces of 32 rows x 16 NF columns: 8 per wave
                const float* __restrict__ resid = p.resid;
                float* __restrict__ out = (float*)p.out;
                // The residual of piece q + RD is requested before piece q is combined and stored (one wave per SIMD, nothing else
                // hides the latency); only tiles with a column edge come here (see the buffer form above for what paces it).
                constexpr int RD = 1;
                f32x4 rs[RD + 1][MI][NF];
                auto load_piece = [&](int q, f32x4 (&dst)[MI][NF]) {
                    const int h = q & 1, sg = q >> 1;
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const size_t ro = (size_t)min(mrow0 + (sg * MI + i) * 16 + fr, p.M - 1) * p.ldo;
#pragma unroll
                        for (int j = 0; j < NF; ++j) {
                            dst[i][j] = *reinterpret_cast<const f32x4*>(resid + ro + min(nb0 + (h * NF + j) * 16 + fq * 4, p.N - 4));
                        }
                    }
                };
#pragma unroll
                for (int q = 0; q < RD; ++q) load_piece(q, rs[q]);
                f32x4 bias[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    bias[j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + min(nb0 + j * 16 + fq * 4, p.N - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int h = q & 1, sg = q >> 1;
                    if (q + RD < 8) load_piece(q + RD, rs[(q + RD) % (RD + 1)]);
                    f32x4 acc[MI][NF];
#define W_RD(n, R, C0, C1, C2, C3) \
                    if (((n) & 7) < NJ && ((n) & 7) / NF == h && ((n) >> 4) == sg) W_READ(acc[((n) >> 3) & 1][((n) & 7) % NF], C0, C1, C2, C3);
                    W_FOR_EACH_ACC(W_RD)
#undef W_RD
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int m = mrow0 + (sg * MI + i) * 16 + fr;
#pragma unroll
                        for (int j = 0; j < NF; ++j) {
                            const int n = nb0 + (h * NF + j) * 16 + fq * 4;
                            if (m < p.M && n < p.N)
                                *reinterpret_cast<f32x4*>(out + (size_t)m * p.ldo + n) = rs[q % (RD + 1)][i][j] + p.alpha * (acc[i][j] + bias[h * NF + j]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (!plain_resid && !buf_tile) {
            // (lookup-free bf16 epilogues: the tile's output rows behind one descriptor, the wave's bias columns loaded once —
            // an empty descriptor without a bias, ending at N: see gemm_epilogue_tile_lds_plain_buf)
            const unsigned ldo2 = uni32((unsigned)p.ldo * 2u);
            const auto ors16 = __builtin_amdgcn_make_buffer_rsrc(uni_ptr((const char*)p.out + (size_t)m0 * p.ldo * 2), 0,
                                                                 uni32((unsigned)min(G256_BM, p.M - m0) * ldo2), 0x00020000);
            f32x4 biasw[NJ];
            if constexpr (PLAIN && NF == 4 && EPI != EPI_ROPE) {
                const auto brs = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(p.bias), 0, uni32(p.bias ? (unsigned)p.N * 4u : 0u), 0x00020000);
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    biasw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (unsigned)(nb0 + fq * 4) * 4u + j * 64, 0, 0));
            }
            // (row bias: only the bf16 form carries one, and only when a tile cannot wrap around its period — launch_w)
            const bool rb_on = EPI == EPI_BF16 && PLAIN && p.rowbias != nullptr;
            const auto rbrs = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(p.rowbias), 0,
                                                                uni32(rb_on ? (unsigned)p.rowbias_period * (unsigned)p.rowbias_ld * 4u : 0u), 0x00020000);
            const unsigned rb_row0 = rb_on ? uni32((unsigned)(m0 % p.rowbias_period)) : 0u;
            int posw[8];               // EPI_ROPE: the positions of this lane's eight rows (row = wm * 128 + t * 16 + fr), one round trip per tile
#pragma unroll
            for (int t = 0; t < 8; ++t) posw[t] = 0;
            if constexpr (EPI == EPI_ROPE && PLAIN) {
#pragma unroll
                for (int t = 0; t < 8; ++t) posw[t] = p.rope_pos[min(mrow0 + t * 16 + fr, p.M - 1)];
            }
            // ... and their cos / sin rows, a strip group (two rows per lane) ahead of the piece that rotates with them
            f32x4 rt[2][2][4];
            auto load_rope = [&](int sg, f32x4 (&d)[2][4]) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float* tab = p.rope_table + (size_t)posw[sg * 2 + i] * 64 + fq * 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const f32x4*>(tab + j * 16);       // cos 0:16, cos 16:32, sin 0:16, sin 16:32
                }
            };
            if constexpr (EPI == EPI_ROPE && PLAIN) load_rope(0, rt[0]);
            (void)ors16; (void)ldo2; (void)biasw; (void)rbrs; (void)rb_row0; (void)posw; (void)rt;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                constexpr int MI = 2;
                const int h = q & 1, sg = q >> 1;       // column half, 32-row strip group
                f32x4 acc[MI][NF];                      // [16-row strip][16-column fragment]
#define W_RD(n, R, C0, C1, C2, C3) \
                if (((n) & 7) < NJ && ((n) & 7) / NF == h && ((n) >> 4) == sg) W_READ(acc[((n) >> 3) & 1][((n) & 7) % NF], C0, C1, C2, C3);
                W_FOR_EACH_ACC(W_RD)
#undef W_RD
                const int mr = mrow0 + sg * 32, nb = nb0 + h * (16 * NF);
                bool done = false;
                if constexpr (EPI == EPI_F32) {
                    // fp32 outputs without per-row lookups (incl. the split-K partial products: split s writes to its
                    // own plane, the bias goes with split 0): bias requested once per piece, predicated vector stores.
                    // The shared per-row epilogue (a branch nest and a dependent bias load per fragment) took 16.9 us
                    // of a 34 us launch on the decoder's o-projection shape.
                    if (!p.rowmap && !p.rowbias) {
                        float* __restrict__ out = (float*)p.out + (size_t)split * p.split_stride;
                        const float* bp = split == 0 ? p.bias : nullptr;
                        f32x4 bias[NF];
#pragma unroll
                        for (int j = 0; j < NF; ++j)
                            bias[j] = bp ? *reinterpret_cast<const f32x4*>(bp + min(nb + j * 16 + fq * 4, p.N - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int i = 0; i < MI; ++i) {
                            const int m = mr + i * 16 + fr;
#pragma unroll
                            for (int j = 0; j < NF; ++j) {
                                const int n = nb + j * 16 + fq * 4;
                                if (m < p.M && n < p.N) *reinterpret_cast<f32x4*>(out + (size_t)m * p.ldo + n) = acc[i][j] + bias[j];
                            }
                        }
                        done = true;
                    } else if (ks > 1) {
                        GemmArgs ps = p;
                        ps.out = (float*)p.out + (size_t)split * p.split_stride;
                        if (split > 0) ps.bias = nullptr;
#pragma unroll
                        for (int i = 0; i < MI; ++i) gemm_epilogue_row<EPI_F32, NF>(acc[i], ps, mr + i * 16 + fr, nb, fq);
                        done = true;
                    }
                }
                if constexpr (EPI == EPI_RESID) {       // (row-mapped residual outputs)
                    gemm_epilogue_resid_tile<MI, NF, 2>(acc, p, mr + fr, nb, fq);
                    done = true;
                }
                if constexpr (NF == 4 && (EPI == EPI_BF16 || EPI == EPI_GELU || EPI == EPI_SWIGLU || EPI == EPI_ROPE)) {
                    if constexpr (PLAIN) {
                        f32x4 b4[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) b4[j] = EPI == EPI_ROPE ? f32x4{0.f, 0.f, 0.f, 0.f} : biasw[h * NF + j];
                        if constexpr (EPI == EPI_ROPE) { if (h == 0 && sg + 1 < 4) load_rope(sg + 1, rt[(sg + 1) & 1]); }
                        gemm_epilogue_tile_lds_plain_buf<EPI, MI>(acc, b4, p, ors16, (unsigned)(wm * 128 + sg * 32) * ldo2, ldo2, rt[sg & 1], nb, lane, wl0,
                                                                  rbrs, (rb_row0 + (unsigned)(wm * 128 + sg * 32)) * (unsigned)p.rowbias_ld * 4u);
                        done = true;
                    } else if ((p.N & 7) == 0 && (p.ldo & 7) == 0) {
                        gemm_epilogue_tile_lds<EPI, MI>(acc, p, mr, nb, lane, wl0);
                        done = true;
                    }
                }
                if constexpr (!PLAIN) {
                    if (!done) {
#pragma unroll
                        for (int i = 0; i < MI; ++i) gemm_epilogue_row<EPI, NF>(acc[i], p, mr + i * 16 + fr, nb, fq);
                    }
                }
                // (lookup-free pieces may overlap in pairs; they share the staging buffer, whose accesses hipcc keeps
                // in program order)
                if (!PLAIN || (q & 1)) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#ifdef VR_W_TIMING
    if (threadIdx.x == 0 && p.rope_table) {
        const unsigned long long tm3 = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tm4 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* d = (unsigned long long*)p.rope_table + (size_t)blockIdx.x * 16;
        d[0] = tm0; d[1] = tm1; d[2] = tm2; d[3] = tm3; d[4] = tm4;
        d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID
        d[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // XCC_ID
        d[7] = cy2 - cy1;                                         // shader clocks spent in the K-loop
    }
#endif
}
