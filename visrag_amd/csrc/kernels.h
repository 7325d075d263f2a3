// Launcher prototypes of the gfx950 kernels (host side of libvisrag_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vr {

// ---- GEMM (gemm.hip) ---------------------------------------------------------------------
enum GemmEpilogue { EPI_BF16 = 0, EPI_GELU = 1, EPI_F32 = 2, EPI_RESID = 3, EPI_SWIGLU = 4, EPI_ROPE = 5 };
// GLDS: 128x128 4-wave tile; 256IL: 256x256 8-wave tile (needs W and A readable up to the next
// multiple of 256 rows); 192: 256x192 tile for N % 192 == 0; AUTO picks by shape.  (The gaps in the ids are
// retired round-1 experiment variants: DESIGN.md ledger row 5.)
// 128W_192 / 128W_256: the half-height one-wave tiles of gemm128w.hip (few rows: the decoder's o / down projections).
enum GemmVariant { GEMM_VARIANT_GLDS = 0, GEMM_VARIANT_AUTO = 3, GEMM_VARIANT_192 = 7, GEMM_VARIANT_256IL = 9, GEMM_VARIANT_256W = 12,
                   GEMM_VARIANT_192W = 13, GEMM_VARIANT_128W_192 = 14, GEMM_VARIANT_128W_256 = 15 };

struct GemmArgs {
    const void* A; int lda;          // bf16 [M_pad][lda]
    const void* W; int ldw;          // bf16 [N][ldw]   (nn.Linear layout: [out][in])
    int M, N, K;                     // N % 128 == 0, K % 64 == 0
    const float* bias;               // f32 [N] or null
    const float* resid;              // f32 [M][ldo] (EPI_RESID)
    float alpha;                     // EPI_RESID scale
    void* out; int ldo;
    const int* rowmap;               // optional: output row of input row m (-1 = drop)
    const float* rowbias;            // optional: f32 [period][rowbias_ld], added for n < rowbias_cols
    int rowbias_period, rowbias_ld, rowbias_cols;
    const int* rope_pos;             // EPI_ROPE: position of row m
    const float* rope_table;         // f32 [max_pos][64] = cos[32] | sin[32]
    int rope_cols;                   // columns < rope_cols are rotated (q and k), rest copied (v)
    int raster_gm;                   // 256-tile kernels: m-tiles per raster group (0 = choose by W size)
    int ksplit;                      // 256-tile kernels, EPI_F32 only: split K over ksplit workgroups per tile;
    size_t split_stride;             //   split s writes its partial product to out + s * split_stride (elements)
    float col_scale; int col_scale_n;   // optional, bf16-output plain epilogue (EPI_BF16, no row map / row bias): columns n < col_scale_n
                                     // (a multiple of 64) leave as bf16((acc + bias) * col_scale) — the ViT's q columns carry
                                     // head_dim^-0.5 * log2(e) into the attention kernel with ONE rounding (attention_w.hip)
    const int* m_dev; int m_sub;     // optional, GEMM_VARIANT_256IL only: rows < *m_dev - m_sub exist (a row count known on the
                                     // device only: tiles at or past it leave at once — the search's band pass)
};
hipError_t launch_gemm(const GemmArgs& a, int epilogue, int variant, hipStream_t s);
// 256x192 tile (gemm192.hip): N % 192 == 0, epilogues BF16 / GELU / F32 / RESID only
hipError_t launch_gemm192(const GemmArgs& a, int epilogue, hipStream_t s);
// 256x256 tile, one wave per SIMD (gemm256w.hip): same contract as the 256-tile path of launch_gemm
hipError_t launch_gemm256w(const GemmArgs& a, int epilogue, hipStream_t s);
bool gemm256w_fits(const GemmArgs& a, int tile_cols);   // its 32-bit LDS-DMA offsets cover both operands
// its 256x192 form: N % 192 == 0, EPI_RESID and EPI_F32 (incl. ksplit)
hipError_t launch_gemm192w(const GemmArgs& a, int epilogue, hipStream_t s);
// 128 x 192 / 128 x 256 tile, one wave per SIMD, three LDS stages (gemm128w.hip): N % tile_cols == 0, no row map / row bias /
// split K; tile_cols 192: EPI_RESID, 256: EPI_RESID / BF16 / GELU / SWIGLU
hipError_t launch_gemm128w(const GemmArgs& a, int epilogue, int tile_cols, hipStream_t s);
bool gemm128w_fits(const GemmArgs& a, int tile_cols);

// M <= 16 rows (gemm_skinny.hip): the decode step's weight streamer; fp32 planes out[split][M][ldo] (+ bias with split 0)
// swiglu (ksplit 1, 16-row [gate | up] interleaved W): out = bf16 act [M][ldo] = silu(gate) * up instead of a plane
// combine (M = 1, K = heads * 128): the A row is merged on the fly from the decode step's partial attention rows
// (bf16 part[S][ldp], lse f32 [S][heads]: see launch_attn_combine) — no separate merge launch
// part / lse come from the decode attention with the `group` query heads of a KV head as ROWS: range t, query head hq =
// hkv * group + g  ->  part[((t * group + g) * (heads / group) + hkv) * 128 + d],  lse[(t * group + g) * (heads / group) + hkv]
struct SkinnyCombine { const void* part; const float* lse; int S; int heads; int group; const int* S_dev; };
hipError_t launch_gemm_skinny(const GemmArgs& a, hipStream_t s, bool swiglu = false, const struct SkinnyCombine* combine = nullptr);
// act = silu(gate) * up from the fp32 planes of a gate/up GEMM over 16-row interleaved weights
hipError_t launch_swiglu_sum(const float* parts, int n_parts, size_t plane_stride, int ldp, int M, int I, void* act, int lda,
                             hipStream_t s);

// ---- norms (norm.hip) --------------------------------------------------------------------
// x f32 [rows][ldx] (dim used) -> bf16 [rows][ldo]; columns [dim, ldo) are written as zero.
hipError_t launch_layernorm(const float* x, int rows, int dim, int ldx, const float* w, const float* b,
                            float eps, void* out, int ldo, hipStream_t s);
// x += alpha * sum_s partial[s] (fp32, fixed order: deterministic), written back; then RMSNorm -> out
// (out == null: accumulate only).  Closes a split-K GEMM (GemmArgs::ksplit) without a reduction pass.
hipError_t launch_rmsnorm_accum(float* x, int rows, int dim, int ldx, const float* partial, int nsplit,
                                size_t split_stride, int ldp, float alpha, const float* w, float eps, void* out,
                                int ldo, hipStream_t s);
hipError_t launch_rmsnorm(const float* x, int rows, int dim, int ldx, const float* w, float eps,
                          void* out, int ldo, hipStream_t s);

// ---- attention (attention.hip) -----------------------------------------------------------
struct AttnArgs {
    const void* q; int ldq;          // bf16, row stride in elements; head h at column h*head_dim
    const void* k; int ldk;
    const void* v; int ldv;
    void* out; int ldo;              // bf16 [rows_q][ldo], head h at column h*head_dim
    const int* cu_q;                 // [B+1] row ranges of q/out (ignored for q when q_shared)
    const int* cu_kv;                // [B+1] row ranges of k/v
    int B, heads, head_dim, max_q;
    int causal, q_shared;            // q_shared: q rows [0,max_q) are the same for every batch item
    float scale;
    int kv_group;                    // grouped-query attention: query head h reads K/V head h / kv_group (0 or 1: one each)
    const int* kv_end;               // optional [B]: item b's K/V rows are [cu_kv[b], kv_end[b]) instead of [cu_kv[b], cu_kv[b+1])
                                     // (ranges of DIFFERENT caches: batched decode)
    const int* q_in_rows;            // optional [B]: first q row of item b (overrides cu_q / q_shared for READING q; out rows stay cu_q)
    int q_head_stride;               // elements between the heads of a q row (0: head_dim) — lets the query HEADS of a
                                     // grouped-query group be handed in as the ROWS of one tile (decode: K/V read once per group)
    int q_prescaled;                 // the q rows already carry scale * log2(e) (GemmArgs::col_scale): the kernels apply no scale
    float* lse;                      // optional f32 [rows_q][heads]: log2 of the row's softmax denominator (with the running
                                     // max folded in) — lets a caller merge attention over separately processed KV ranges
};
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);

// ---- EVisRAG generator (gen_kernels.hip) ---------------------------------------------------
constexpr int GEN_ATT_SPLITS = 16;  // most KV ranges one decode step's attention is cut into
// What a decode step needs beyond the caches, resident on the device so that consecutive steps need no host round trip
// (gen.hip: host-driven steps upload the first five words; free-running steps advance them on the device).
struct GenState {
    int token;                      // the token this step appends
    int pos[3];                     // its temporal / height / width position
    int len;                        // KV-cache rows in use before this step
    int step;                       // index of the next sampled token (selects the sampling noise)
    int splits;                     // KV ranges of this step's attention (decode_begin_kernel)
    int pad;
    int cu_q[GEN_ATT_SPLITS + 1], cu_kv[GEN_ATT_SPLITS + 1];
};
// one vg_decode_batch step (up to 16 sequences): built on the host, uploaded once per step
struct GenBatch {
    int token[16];
    int pos[3][16];                                  // [component][row]
    int cache_row[16];                               // cache row the row's new token goes to: slot * max_len + len
    int splits[16];                                  // KV ranges of the row's attention
    int cu_q[16 * GEN_ATT_SPLITS + 1];               // output rows of item (row, range): (row * SPLITS + range) * group
    int kv_lo[16 * GEN_ATT_SPLITS], kv_hi[16 * GEN_ATT_SPLITS];   // the item's cache rows [lo, hi)
    int q_in[16 * GEN_ATT_SPLITS];                   // the item's first q row (128-wide rows: row * heads)
    int sampled[16];                                 // vg_sample_batch: the sampled tokens
};
hipError_t launch_decode_begin(GenState* st, int q_rows, hipStream_t s);
// All decoder layers of ONE decode step (one sequence) as one persistent launch, a workgroup per CU (gen_persist.hip).
struct PersistLayer {
    const void *wqkv, *wo, *wgu, *wd;       // bf16 [n_pad][k_pad] (gate / up rows interleaved in blocks of 16: EPI_SWIGLU's layout)
    const float* bqkv;                      // f32 [QKV] or null
    const float *g1, *g2;                   // input / post-attention norm weights of this layer
    void *kc, *vc;                          // the sequence's K / V cache of this layer: bf16 [rows][KVD]
};
struct PersistArgs {
    const PersistLayer* layers; int n_layers;
    int E, QKV, QD, KVD, H, KV, Ip, N2;     // hidden, q|k|v columns, q columns, KV columns, heads, KV heads, act row length, gate|up rows (padded)
    int ldw_qkv, ldw_o, ldw_gu, ldw_d;      // row lengths (k_pad) of the four weight matrices
    int ks_qkv, ks_o, ks_d;                 // K splits (gate|up is unsplit: SwiGLU in its epilogue)
    float eps;
    const float* g_final;                   // the model's final norm weight
    float* h; void* xn;                     // residual row f32 [E]; in: the first layer's normalised row (bf16), out: the lm_head's
    float* planes; void* act; void* attp; float* lse; float* ss;   // split-K planes, SwiGLU row, partial attention rows + lse, 16 row sums of squares
    const GenState* st; const float* inv_freq; int sec_t, sec_h;
    unsigned long long* sync;               // 128 words, zeroed once (barrier counters live across launches)
    unsigned* abort_host;                   // host-mapped flag: a barrier timed out
};
hipError_t launch_decode_persist(const PersistArgs& a, int grid, hipStream_t s);
int decode_persist_occupancy();             // workgroups per CU the kernel fits (0: never launch it)
   // q_rows: query rows per KV range (the GQA group size)
// multimodal RoPE on q (in place into q_out) and k (into the K cache at rows cache_row0 + t), v copied into the V
// cache; head_dim 128; pos3 = [3][pos_stride] ints (temporal, height, width); inv_freq f32 [64]; source = bf16 qkv rows
// or (parts != null) fp32 split-K planes + bias; cu_kv (optional) receives {0, cache_row0 + T}
hipError_t launch_mrope_cache(const void* src_bf16, const float* parts, int n_parts, size_t plane_stride, const float* bias,
                              int ld, int T, int H, int KV, const int* pos3, int pos_stride, int sec_t, int sec_h,
                              const float* inv_freq, void* q_out, int ldq, void* k_cache, void* v_cache, int ld_cache,
                              int cache_row0, int* cu_kv, hipStream_t s, const int* row0_dev = nullptr,   // row0_dev: cache_row0 read on the device
                              const int* cache_rows = nullptr);  // cache_rows [T] (device): row t goes to cache row cache_rows[t] (batched decode)
// vision tower: cs f32 [T][64][2] = (cos, sin)((p < sec_h ? pos_h : pos_w)[t] * freq[p]); then q / k head slots (2 * hh
// wide, pairs (p, p + hh), hh <= 64) of bf16 qkv rows rotated in place, slots [0, n_slots) at columns slot * 2 * hh
hipError_t launch_rope2d_table(const int* pos_h, const int* pos_w, int T, int sec_h, const float* freq, void* cs, hipStream_t s);
hipError_t launch_rope2d_inplace(void* qkv, int ld, int T, int n_slots, int hh, const void* cs, hipStream_t s);
hipError_t launch_mark_seen(const int* ids, int n, unsigned* seen, int vocab, hipStream_t s);
// repetition penalty over the seen ids, temperature sampling (Gumbel-max; 0 = argmax), marks the chosen token seen.
// st (optional): the token also goes to st->token; step_from_state: the noise index is st->step; advance: the step's
// bookkeeping (len, step, positions + 1) happens here, on the device
hipError_t launch_sample(const float* logits, int vocab, unsigned* seen, float penalty, float temperature,
                         unsigned long long seed, unsigned step, int* token_out, unsigned long long* scratch64, hipStream_t s,
                         GenState* st = nullptr, int step_from_state = 0, int advance = 0);
hipError_t launch_scatter_rows(const float* src, const int* row_idx, int n, int dim, float* dst, int ld, hipStream_t s);
// dst[i][:] = bf16(src[row_idx[i]][0..dim)), zero up to ld (src rows are dense: stride dim)
hipError_t launch_gather_rows_bf16(const float* src, const int* row_idx, int n, int dim, void* dst, int ld, hipStream_t s);
// dst[i][:] = bf16 patch row (c, t, y, x) of patch (ph[i], pw[i]) of u8 HWC page img[i], (u8 / 255 - mean) / std; zero up to ld
hipError_t launch_patch_rows_u8(const uint8_t* const* pages, const int* page_w, const int* img, const int* ph, const int* pw, int n,
                                int P, int tp, const float* mean3, const float* std3, void* dst, int ld, hipStream_t s);
// out[h*128 + d] = sum_s 2^(lse[s][h] - max) * part[s][h*128 + d] / sum_s 2^(lse[s][h] - max): merges the S partial
// attention rows (bf16 [S][ldp], each normalised over its own KV range) of one decode step
hipError_t launch_attn_combine(const void* part, const float* lse, int S, int heads, int group, void* out, hipStream_t s,
                               const int* S_dev = nullptr,       // S_dev: S read on the device; layout: see SkinnyCombine
                               int n_rows = 1, int ld_out = 0);  // batched decode: row r's ranges start at range r * GEN_ATT_SPLITS,
                                                                 // its count is S_dev[r], its output row is out + r * ld_out

// ---- elementwise / gather / pooling (misc.hip) ---------------------------------------------
// Fused ToTensor/Normalize + patch-embed conv (patch_embed.hip): g carries the PERMUTED weight (k = ky*3P + kx*3 + c),
// bias, the resampled pos-embed as rowbias (period = patches per image), out fp32 [M][ldo], M = n_imgs * patches.
hipError_t launch_patch_embed(const uint8_t* const* imgs, int n_imgs, int H, int W, int P, const GemmArgs& g, int Kreal,
                              hipStream_t s);
// h[t][:] = table[ids[t]][:] * scale   (bf16 table -> f32 rows)
hipError_t launch_embed_gather(const int* ids, int T, const void* table, int dim, float scale,
                               float* out, hipStream_t s);
// final RMSNorm + position-weighted mean pool + L2 normalise: one embedding per sequence.
hipError_t launch_pool(const float* h, const int* seq_offsets, int B, int dim, const float* norm_w,
                       float eps, float* out, float* tap_hidden, hipStream_t s, int mode = 0);   // mode: VR_POOL_*
hipError_t launch_f32_to_bf16(const float* in, void* out, size_t n, hipStream_t s);
hipError_t launch_f32_to_bf16_pad(const float* in, void* out, size_t n, size_t n_total, hipStream_t s,
                                  int* zero_word = nullptr);   // + zero tail (+ TWO ints cleared)
// split x (f32) into hi + lo bf16 parts (x ~= hi + lo to ~16 bits of mantissa)
hipError_t launch_split_bf16(const float* in, void* hi, void* lo, size_t n, hipStream_t s);
hipError_t launch_iota_pos(const int* seq_offsets, int B, int* pos, hipStream_t s);
hipError_t launch_any_nonzero16(const void* p, size_t n, int* flag, hipStream_t s);   // *flag |= any bf16 word != +-0

// ---- split-precision text path (hp_text.hip): fp32 glue between hi + lo bf16 GEMMs ---------------
hipError_t launch_rmsnorm_split(const float* x, int rows, int dim, const float* w, float eps, void* hi, void* lo, hipStream_t s);
hipError_t launch_rope_f32(float* qkv, int T, int ld, int rope_cols, const int* pos, const float* table, hipStream_t s);
hipError_t launch_attn_f32(const float* qkv, int ld, int E, const int* seq_of, const int* seq_offsets, int T, int heads,
                           float scale, float* out, hipStream_t s);
hipError_t launch_swiglu_split(const float* gu, int T, int ld_gu, int I, int ld_act, void* hi, void* lo, hipStream_t s);
hipError_t launch_embed_gather_hp(const int* ids, int T, const void* table_hi, const void* table_lo, int dim, float scale,
                                  float* out, hipStream_t s);
hipError_t launch_seq_of(const int* seq_offsets, int B, int* seq_of, hipStream_t s);
// out = (accumulate ? out : 0) + alpha * sum of n_parts fp32 planes [T][ldp] (columns < N), fixed order
hipError_t launch_planes_sum(const float* parts, int n_parts, size_t stride, int ldp, int T, int N, float* out, int ldo, float alpha,
                             bool accumulate, hipStream_t s);

// ---- synthetic page images (synth.hip; bench / test support): pages first .. first + n - 1 of visrag_amd/synth.py's corpus
hipError_t launch_synth_pages(uint8_t* out, int n, int size, long long seed, long long first, hipStream_t s);

// ---- PIL-exact bicubic resize (resize.hip) ------------------------------------------------------
} // namespace vr
#include <vector>
namespace vr {
int resize_coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& kk);   // returns ksize
hipError_t launch_resize_h(const uint8_t* in, int in_w, int rows, uint8_t* out, int out_w, const int* bounds,
                           const int* kk, int ksize, hipStream_t s);
hipError_t launch_resize_v(const uint8_t* in, int w, uint8_t* out, int out_h, const int* bounds, const int* kk,
                           int ksize, hipStream_t s);

// ---- search (search.hip) -------------------------------------------------------------------
struct SearchArgs {
    const void* index_bf16;          // [n_docs_pad][dim] bf16
    const float* index_f32;          // [n_docs][dim] f32
    int64_t n_docs;
    int dim;
    const void* q_bf16;              // [nq_pad][dim] bf16
    int convert_q;                   // streaming kernel only: q_bf16 has NOT been filled — every workgroup converts the queries
                                     // from q_f32 itself, workgroup 0 writes q_bf16 rows [0, nq) and clears the two flag counters
    const float* q_f32;              // [nq][dim]
    int nq, k;
    float* cand_scores; int* cand_ids; int n_chunks;   // workspace [nq_pad][n_chunks][KP]
    float* out_scores; int64_t* out_ids;               // [nq][k]
    float* thr_init;                                   // workspace [nq_pad] or null (no pre-pass)
    int pre_own_chunks;                                // 0, or the number of list chunks at the END of n_chunks that belong to the threshold
                                                       // pre-pass (search_prepass_owned): its sampled 256-row tiles are scored ONCE — the pre-pass
                                                       // keeps every score, appends the rows >= the threshold to those lists itself, and the
                                                       // sweep (n_chunks - pre_own_chunks workgroup chunks) skips the sampled tiles
    float* thr_cert;                                   // workspace [nq_pad] (pre_own_chunks > 0): what the lists are complete down to — the
                                                       // threshold, or +inf for a query whose pre-pass list overflowed (exact ties)
    unsigned long long* cand_keys;                     // 256-tile sweep scratch [nq_pad256][n_chunks][2][64] or null
    float* score_rows; size_t ld_scores;               // streaming search (nq <= 16) only: the sweep writes EVERY bf16-MFMA score,
                                                       // [16][ld_scores] (ld_scores % 256 == 0, >= n_docs), and a query its merge cannot
                                                       // certify is redone in place by its own merge workgroup (search_band.h)
    int exact_follows;               // streaming search with score_rows: 1 = the exact fp32 pass IS launched behind this merge — a query whose band
                                     // exceeds search_band_max() rows goes on the flag2 list instead of being walked by its one workgroup
    int* huge_seen;                  // host-visible word (or null): set when a merge workgroup had to walk such a band itself (exact_follows == 0);
                                     // the engine launches the exact pass behind the index's streaming searches from then on
    // ---- certification of the candidate selection (search_common.h: certify_tail)
    const float* thr_used;           // thresholds the sweep STARTED from (set by the launcher; null: none)
    float eps_rel;                   // >= 0: |bf16-MFMA score - fp32 score| <= eps_rel * |q| * dmax (the caller's model);
                                     // < 0: no certification; eps_data set: ignored, the bound is search_common.h's query_eps
    int eps_data;                    // 1: the default, data-dependent rigorous bound (needs dmax[1] and acc_rel)
    float acc_rel;                   // its accumulation term: (2 dim + 128) 2^-24
    const float* dmax;               // device floats: [0] largest row norm |d|, [1] largest bf16 rounding residual |d - bf16(d)|
    int* flag_count; int* flag_list; // queries whose top-k could not be certified: the band pass (search_band.hip) redoes them
    float* flag_tau;                 // [slot] the flagged query's tau = s_k - eps (-inf: unknown)
    void* flag_q;                    // bf16 [slots][dim]: the flagged queries' bf16 rows, compacted (the band pass's GEMM operand)
    int* flag2_count; int* flag2_list;   // flagged queries whose band holds too many rows: the exact fp32 pass redoes them
    unsigned* stats;                 // [0] certified at once [1] after extended re-scoring [2] flagged [3] uncertified mode
                                     // [4] candidates gathered a second time [5] of the flagged: redone by the exact fp32 pass
    // ---- packed output (multi-GPU exchange format): key = orderable(score) << 32 | ~(row + id_offset); 0 = none
    unsigned long long* out_keys; int64_t id_offset;   // when set, out_scores / out_ids are not written
    hipEvent_t* prof_ev;             // optional [SEARCH_PROF_EVENTS] stage marks recorded on the launch stream
};
constexpr int SEARCH_PROF_EVENTS = 6;   // start | queries converted | thresholds | sweep | merge | exact pass
int search_kprime(int k);            // candidates kept per (query, chunk); 0 if k unsupported
int search_num_chunks(int64_t n_docs, int nq);         // workgroup chunks of the sweep (over the rows it sweeps: see search_prepass_owned)
// 8 when the threshold pre-pass OWNS its sample (the 256-tile sweep on the one-wave kernel, >= 128 index tiles), else 0: the caller
// adds it to n_chunks (list chunks) and sets SearchArgs::pre_own_chunks / thr_cert
int search_prepass_owned(int64_t n_docs, int nq, int dim);
constexpr int SEARCH_PRE_SPOTS = 16;    // sampled 256-row tiles of the owning pre-pass: tile k * S + S - 1, S = index tiles / 16
int search_prepass_floats();         // floats of cand_scores per (padded) query the threshold pre-pass needs
hipError_t launch_search(const SearchArgs& a, hipStream_t s);
bool search_uses_stream(int nq, int dim);   // nq <= 16: index streamed through registers (search_small.hip)
int search_stream_chunks();
hipError_t launch_search_stream(const SearchArgs& a, int kp, hipStream_t s);   // sweep only; merge: search_merge_wg_kernel
bool search_uses_256(int nq);         // more than 128 queries: main sweep on the 256^2 tile (search256.hip)
hipError_t launch_sweep256(const SearchArgs& a, int kp, const float* thr, hipStream_t s);
// the same sweep on the one-wave-per-SIMD tile (search256w.hip; dim % 128 == 0): sweep only, launch_sweep256 merges
bool sweep256w_ok(const SearchArgs& a);
hipError_t launch_sweep256w(const SearchArgs& a, int kp, const float* thr, hipStream_t s);   // (honours pre_own_chunks)
// k > 26 (search_bigk.hip): radix select over the block's score rows S + exact re-score; k <= search_bigk_max()
int search_bigk_max();
hipError_t launch_search_bigk(const SearchArgs& a, const float* S, size_t ldS, int q0, int nq_block, hipStream_t s);
hipError_t launch_topk_merge_big(const float* scores, const int64_t* ids, int n_parts, int nq, int k,
                                 float* out_scores, int64_t* out_ids, hipStream_t s);   // n_parts * k <= 8192
hipError_t launch_topk_merge(const float* scores, const int64_t* ids, int n_parts, int nq, int k,
                             float* out_scores, int64_t* out_ids, hipStream_t s);
// the same merge over packed keys [n_parts][nq][k] (SearchArgs::out_keys format; ids are global already)
hipError_t launch_topk_merge_keys(const unsigned long long* keys, int n_parts, int nq, int k, float* out_scores,
                                  int64_t* out_ids, hipStream_t s);      // n_parts * k <= 8192
// ---- exact fp32 pass for the flagged queries (search_exact.hip)
// S[slot][doc] = fp32 dot(query flag_list[slot], row doc) for slot < *flag_count, in the summation order of the
// re-scoring (search_common.h: dot_lane) — every workgroup leaves at once when nothing is flagged
// entries [sub, sub + max_slots) of the list are this launch's slots 0.. (a pass over a bounded score buffer)
hipError_t launch_exact_scores(const float* index_f32, int64_t n_docs, int dim, const float* q_f32, const int* flag_list,
                               const int* flag_count, int sub, int max_slots, float* S, size_t ldS, hipStream_t s);
// top-k of the flagged queries from their exact score rows (radix select + re-score + sort): overwrites their outputs
// (a.flag_count / a.flag_list: the list the scores were made for)
hipError_t launch_exact_select(const SearchArgs& a, const float* S, size_t ldS, int sub, int max_slots, hipStream_t s);
// ---- band pass for the flagged queries (search_band.hip): S[slot][doc] = bf16-MFMA scores of flagged query
// flag_list[sub + slot] (GEMM over a.flag_q); every row with a score >= its tau re-scored in fp32, top k emitted; a band of
// more than search_band_max() rows goes on the flag2 list (exact fp32 pass)
int search_band_max();
hipError_t launch_band_select(const SearchArgs& a, const float* S, size_t ldS, int sub, int max_slots, hipStream_t s);
// dmax[0] = max(dmax[0], |row|), dmax[1] = max(dmax[1], |row - bf16(row)|) over n rows (vr_index_add)
hipError_t launch_row_norm_max(const float* rows, int64_t n, int dim, float* dmax, hipStream_t s);
float search_default_eps_rel(int dim);      // worst-case relative bound (a caller's yardstick; the default bound is data-dependent)
float search_acc_rel(int dim);

}  // namespace vr
