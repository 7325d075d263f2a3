/*
 * visrag_hip.h — C ABI of libvisrag_hip.so: the MI355X (gfx950) native VisRAG-Ret
 * corpus-embedding + retrieval hot path.
 *
 * The reference (OpenBMB/VisRAG) has NO FFI for this path — it is pure Python over
 * PyTorch ops — so this header is new surface.  Each entry point states the reference
 * interface (file:line under /root/reference) whose device math it replaces; the Python
 * adapter in visrag_amd/ keeps the reference's operator API on top of it
 * (INTEGRATION.md shows the ctypes binding a maintainer would add).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; vr_last_error() gives text
 *     (reference convention is Python exceptions: dense_retriever.py:70-71,
 *      inference.py:105-108 — the adapter raises RuntimeError from the status).
 *   - plain pointers and sizes only; no torch types.  "dev" pointers are HIP device
 *     pointers (e.g. tensor.data_ptr() of a torch-ROCm tensor used as a container).
 *   - the library owns device weights, workspace and the HBM-resident index; the caller
 *     owns every input and output buffer.  No cross-boundary frees.
 *   - a handle is NOT thread-safe: one handle per process / GPU (the reference runs one
 *     process per GPU under torchrun, eval.sh:48).  `stream` is a hipStream_t (NULL = the
 *     default stream) so torch containers stay ordered with the kernels.
 */
#ifndef VISRAG_HIP_H
#define VISRAG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VR_OK 0
#define VR_ERR_INVALID 1   /* bad argument / shape / missing weight */
#define VR_ERR_HIP 2       /* HIP runtime failure */
#define VR_ERR_STATE 3     /* call order (e.g. encode before weights are finalised) */
#define VR_ERR_CAPACITY 4  /* workspace / index capacity exceeded */

#define VR_DTYPE_F32 0
#define VR_DTYPE_BF16 1

typedef struct vr_model_s* vr_model_t;
typedef struct vr_index_s* vr_index_t;

/* Model dimensions (visrag_amd/config.py; SURVEY.md section 8 "dimension provenance"). */
typedef struct vr_config {
    int32_t patch_size;        /* 14 */
    int32_t vit_dim;           /* 1152 */
    int32_t vit_depth;         /* 26 (27 in the checkpoint, last dropped: modeling_minicpmv.py:70-71) */
    int32_t vit_heads;         /* 16  -> head_dim must be 72 */
    int32_t vit_hidden;        /* 4304 = int(1152*3.7362) */
    int32_t vit_pos_grid;      /* 27  (learned pos-embed is 27x27) */
    float   vit_ln_eps;        /* 1e-6 */
    int32_t query_num;         /* 64 */
    float   resampler_ln_eps;  /* 1e-6 */
    int32_t hidden_size;       /* 2304 -> resampler heads = hidden/128, head_dim 128 */
    int32_t num_layers;        /* 40 */
    int32_t num_heads;         /* 36  -> head_dim must be 64 */
    int32_t intermediate_size; /* 5760 */
    int32_t vocab_size;        /* 122753 */
    float   rms_norm_eps;      /* 1e-5 */
    float   rope_theta;        /* 1e4 */
    float   scale_emb;         /* 12 */
    float   residual_scale;    /* scale_depth / sqrt(num_layers) = 1.4/sqrt(40) */
    int32_t max_images;        /* workspace: image slices per ViT pass (chunk size) */
    int32_t max_patches;       /* workspace: patches per image slice (1024 for 448x448; <=1064 sliced) */
    int32_t max_tokens;        /* workspace: packed decoder tokens per vr_encode call */
    int32_t max_seqs;          /* workspace: sequences per vr_encode call */
    int32_t text_split_precision; /* 1: token-only batches (queries, text passages) run the decoder on hi + lo bf16
                                   * splits of activations and fp32 source weights with fp32 glue (fp32-class accuracy:
                                   * the 1e-3 score bar for ~20-token queries); 0: the bf16 path for everything;
                                   * N > 1: only token-only batches whose longest sequence has <= N tokens (512 = the
                                   * reference's query length): longer text passages stay on the bf16 MFMA attention path.
                                   * The route is per vr_encode CALL (a call with image slices runs the bf16 pass for all its
                                   * items); the host adapter (visrag_amd/modeling.py: encode_prepared) therefore groups the
                                   * token-only items of a batch into calls of their own, so that an item's embedding does not
                                   * depend on its batch mates. */
} vr_config_t;

/* ---- library ------------------------------------------------------------------------ */
const char* vr_version(void);
/* Last error text of this thread's most recent failing call. */
const char* vr_last_error(void);
int vr_device_count(int* count);

/* ---- model: replaces VisRAG_Ret.forward + pooling ----------------------------------- */
/* Reference: DRModelForInference.build / from_pretrained
 * (src/openmatch/modeling/dense_retrieval_model.py:233-364). */
int vr_model_create(int device_id, const vr_config_t* cfg, vr_model_t* out);
int vr_model_destroy(vr_model_t m);

/* Load one tensor of the HF state dict by its key (e.g.
 * "vpm.blocks.3.attn.qkv.weight", "resampler.proj", "llm.model.layers.7.mlp.up_proj.weight").
 * `data` is row-major with `shape[ndim]`; dtype VR_DTYPE_F32 or VR_DTYPE_BF16; on_device
 * tells whether `data` is a device pointer.  Unused keys (llm.lm_head.*, vpm.attn_pool.*,
 * vpm.blocks.<depth>.*, resampler.pos_embed, rotary buffers) are accepted and ignored.
 * Reference: PreTrainedModel.from_pretrained via dense_retrieval_model.py:292-318. */
int vr_model_load_weight(vr_model_t m, const char* name, const void* data,
                         const int64_t* shape, int32_t ndim, int32_t dtype, int32_t on_device);
/* Check that every required weight arrived and build derived tables (packed / padded
 * bf16 weights, resampler query projection, RoPE table). */
int vr_model_finalize(vr_model_t m);
/* A second handle on the same (finalised) weights with its own workspace: lets a caller keep two
 * batches in flight on two HIP streams (each vr_encode call is still ordered on the stream it is
 * given).  `src` must outlive the clone; the clone is destroyed with vr_model_destroy. */
int vr_model_clone(vr_model_t src, vr_model_t* out);

/* Encode a batch of items (pages and/or text queries) to unit-norm embeddings.
 *   slices      n_slices pointers to uint8 HWC (RGB) images, all on host or all on device
 *   slice_hw    [n_slices][2] = (H, W), multiples of patch_size
 *   input_ids   [T] packed token ids of all B items (already truncated to max_inp_length)
 *   seq_offsets [B+1] token offsets of the items in `input_ids` (seq_offsets[B] == T)
 *   vision_rows [n_slices*query_num] packed-token row that receives resampler output
 *               (slice s, query j), or -1 to drop it (the reference's scatter_ of image_bound)
 *   out_reps    [B][hidden_size] float32, device or host
 * Replaces: VisRAG_Ret.forward (modeling_visrag_ret.py:86-126) incl. ToTensor/Normalize
 * (modeling_minicpmv.py:84-92), get_vllm_embedding / get_vision_embedding (:95-171),
 * timm forward_features (vision_transformer.py:682-692), Resampler.forward
 * (resampler.py:146-168), MiniCPMModel.forward (modeling_minicpm.py:1147-1304), and the
 * wmean pooling + F.normalize of DRModel.encode (dense_retrieval_model.py:180-184,222-223). */
int vr_encode(vr_model_t m,
              const uint8_t* const* slices, const int32_t* slice_hw, int32_t n_slices,
              int32_t slices_on_device,
              const int32_t* input_ids, const int32_t* seq_offsets, int32_t B,
              const int32_t* vision_rows,
              float* out_reps, int32_t out_on_device, void* stream);

/* vr_encode, and ALSO the last hidden states the pooling reads — the HF-style forward of the reference's second caller:
 * `outputs = model(text=..., image=..., tokenizer=...)` returns `last_hidden_state` [B, L, hidden] + `attention_mask` and the
 * caller pools them itself (visrag_scripts/demo/visrag_pipeline/utils.py:12-32, demo/retriever/demo.py:14-36;
 * BaseModelOutputWithAttentionMask, modeling_visrag_ret.py:123-126).
 *   out_hidden  [B][hidden_len][hidden_size] float32 on the DEVICE: item i's post-norm rows, right-padded with zeros like the
 *               reference's `pad` (modeling_minicpmv.py:440-479); hidden_len >= the longest item. */
int vr_encode_hidden(vr_model_t m,
                     const uint8_t* const* slices, const int32_t* slice_hw, int32_t n_slices,
                     int32_t slices_on_device,
                     const int32_t* input_ids, const int32_t* seq_offsets, int32_t B,
                     const int32_t* vision_rows,
                     float* out_reps, int32_t out_on_device,
                     float* out_hidden, int32_t hidden_len, void* stream);

/* Pooling of the last hidden states (DRModel.encode, dense_retrieval_model.py:172-220), applied after the
 * final RMSNorm and followed by the L2 normalisation (:222-223).  VisRAG-Ret's published setting is wmean
 * (the default).  The reference's drop_wmean / drop_mean / lasttoken_simcse apply dropout in training mode
 * even at inference (a fresh nn.Dropout1d, :187,196; F.dropout(training=True), :215): stochastic there,
 * not offered here. */
#define VR_POOL_WMEAN 0      /* sum_t (t+1) h_t / sum_t (t+1)        :180-184 */
#define VR_POOL_MEAN 1       /* mean over the item's tokens           :204-207 */
#define VR_POOL_LASTTOKEN 2  /* last_token_pool, right padding        :26-34,172-177 */
#define VR_POOL_CLS 3        /* hidden[:, 0]                          :217-218 */
int vr_model_set_pooling(vr_model_t m, int32_t mode);

/* Debug taps for parity tests: copy an internal activation of the LAST vr_encode call to
 * host float32.  name in {"vit_embed","vit_block0","vit_out","resampler_out",
 * "inputs_embeds","dec_layer0","last_hidden"}; rows/cols describe `out`. */
int vr_model_tap(vr_model_t m, const char* name, float* out, int64_t rows, int64_t cols);
/* Enable/disable recording of taps (off by default; costs device copies). */
int vr_model_set_taps(vr_model_t m, int32_t enable);

/* Per-kernel-class HIP-event timing on the launch stream (bench.py roofline fields).
 * Enabling resets the counters.  total_flops is the ALGORITHMIC work of the launches
 * (SURVEY.md section 8d formulas), total_ms the sum of their event-bracketed durations. */
#define VR_PROF_VIT_QKV 0    /* gemm_bf16_kernel<EPI_BF16>  : qkv projection            */
#define VR_PROF_VIT_ATTN 1   /* attention_kernel<72,2>      : ViT self-attention        */
#define VR_PROF_VIT_PROJ 2   /* gemm_bf16_kernel<EPI_RESID> : attn out-projection       */
#define VR_PROF_VIT_FC1 3    /* gemm_bf16_kernel<EPI_GELU>  : MLP fc1 + erf-GELU        */
#define VR_PROF_VIT_FC2 4    /* gemm_bf16_kernel<EPI_RESID> : MLP fc2 + residual        */
#define VR_PROF_RESAMPLER 5  /* whole resampler phase (several kernels)                 */
#define VR_PROF_DECODER 6    /* whole 40-layer decoder phase (several kernels)          */
/* the decoder phase again, split (events between its kernels: read these for shares, VR_PROF_DECODER for the total) */
#define VR_PROF_DEC_QKV 7    /* q|k|v projection + RoPE epilogue                        */
#define VR_PROF_DEC_ATTN 8   /* causal attention over the packed sequences              */
#define VR_PROF_DEC_O 9      /* o projection (split-K planes or residual epilogue)      */
#define VR_PROF_DEC_GU 10    /* gate|up projection + SwiGLU epilogue                    */
#define VR_PROF_DEC_DOWN 11  /* down projection                                         */
#define VR_PROF_DEC_NORM 12  /* RMSNorm passes (incl. the split-K accumulate)           */
#define VR_PROF_CLASSES 13
/* enable: 0 off, 1 the phase classes 0..6, 2 the decoder's sub-phases 7..12 instead (their events sit between the decoder's
 * kernels and would lengthen VR_PROF_DECODER if both were taken in one pass) */
int vr_model_set_profile(vr_model_t m, int32_t enable);
int vr_model_get_profile(vr_model_t m, int32_t cls, double* total_ms, int64_t* launches,
                         double* total_flops);

/* ---- index: replaces torch.matmul + torch.topk over pickle shards -------------------- */
/* Reference: _retrieve_one_shard / distributed_parallel_retrieve
 * (src/openmatch/retriever/dense_retriever.py:13-97); demo answer.py:26-35. */
int vr_index_create(int device_id, int32_t dim, int64_t capacity, vr_index_t* out);
int vr_index_destroy(vr_index_t ix);
int vr_index_reset(vr_index_t ix);
/* Append n rows of float32 [n][dim] (host or device); rows keep their insertion order,
 * the global id of a row is its position. */
int vr_index_add(vr_index_t ix, const float* reps, int64_t n, int32_t on_device, void* stream);
int vr_index_size(vr_index_t ix, int64_t* n);
/* For every query the k best rows by inner product: higher score first, lower row id first
 * among equal scores — the ranking torch.topk sees over the fp32 matmul of
 * dense_retriever.py:28-30.  Scores are fp32 dot products of the fp32 rows.  A bf16 MFMA
 * sweep selects candidates, the survivors are re-scored in fp32, and the selection is
 * CERTIFIED per query: with |bf16 score - fp32 score| <= eps (below) every row whose bf16 score
 * could still reach the fp32 top-k is re-scored too; a query whose candidate lists cannot prove
 * completeness (more near-ties at the k-th score than they hold, or a cluster of near-duplicate
 * pages that overran them) is redone by the band pass — bf16 scores of that query against every
 * row, ALL rows inside the error band re-scored in fp32 — and, if its band holds more than 8192
 * rows, by an exact fp32 pass over the whole index.  The ids are the fp32 ranking's.
 *   queries [nq][dim] float32;  out_scores [nq][k] float32;  out_ids [nq][k] int64
 * (all host or all device per `on_device`).  If the index holds fewer than k rows the
 * tail is filled with score -inf, id -1.  k = 1..1000: up to 26 on the fused sweep (the
 * throughput path; --retrieve_depth 10 in eval.sh), deeper runs on GEMM + radix select. */
int vr_index_search(vr_index_t ix, const float* queries, int32_t nq, int32_t k,
                    float* out_scores, int64_t* out_ids, int32_t on_device, void* stream);
/* The error model of the certification.  Default (NaN restores it): the rigorous bound for the
 * data at hand — with dq = q - bf16(q), dd = d - bf16(d) (round to nearest even: bf16 keeps 8
 * significand bits, unit roundoff 2^-8) and fp32 accumulation,
 *   |bf16 score - fp32 score| <= |dq| max|d| + |q| max|dd| + |dq| max|dd|
 *                                + (2 dim + 128) 2^-24 (|q| + |dq|) (max|d| + max|dd|)
 * by Cauchy-Schwarz; |dq| is measured per query, max|d| and max|dd| over the index rows by
 * vr_index_add (~3.4e-3 for unit vectors at dim 2304; the worst case over all data would be
 * 2^-7 + 2^-16 + ... = 8.1e-3).  eps_rel >= 0: the caller's model eps_rel * |q| * max|d| instead;
 * eps_rel < 0: certification off (the bf16 top-(k+6) re-scored, as in rounds 1-2: tolerance-exact). */
int vr_index_set_search_eps(vr_index_t ix, float eps_rel);
/* out4 = {max|d|, max|d - bf16(d)| over the rows added so far, the accumulation term
 * (2 dim + 128) 2^-24, the worst-case relative bound 2^-7 + 2^-16 + accumulation}. */
int vr_index_error_model(vr_index_t ix, float* out4);
/* Queries counted since the last reset: out6[0..5] = {certified at once, certified after extended
 * re-scoring, flagged (redone by the band pass), searched with certification off, of the second
 * group: those whose candidates had to be gathered a second time, of the flagged: those whose band
 * exceeded 8192 rows — redone by the exact fp32 pass, or (the first such queries of an index: the exact
 * pass is only launched behind an index that has shown one) walked by one workgroup}. */
int vr_index_search_stats(vr_index_t ix, int64_t* out6, int32_t reset);
/* How a search of `nq` queries (k <= 26) over the rows added so far would be laid out: out5 = {list chunks per query,
 * of them the chunks that belong to the threshold pre-pass (0, or 8 when the pre-pass OWNS its sample: its 16 sampled 256-row
 * tiles are scored once, their survivors kept in lists of their own, and the sweep skips them), workgroup chunks of the
 * sweep, index tiles (256 rows) per workgroup chunk, whether the exact fp32 pass is launched behind the search (1 once the
 * index has met a band beyond 8192 rows; vr_index_reset clears it)}.  Introspection only (tests, bench.py's `search.plan`). */
int vr_index_search_plan(vr_index_t ix, int32_t nq, int32_t* out5);
/* Per-stage HIP-event times of vr_index_search (k <= 26), summed over calls since enabling:
 * ms5 = {query conversion, threshold pre-pass, sweep, merge + re-scoring, band + exact pass}.
 * While enabled every call ends with an event synchronisation. */
int vr_index_set_search_profile(vr_index_t ix, int32_t enable);
int vr_index_get_search_profile(vr_index_t ix, double* ms5, int64_t* calls);
/* The same search with the result packed for the multi-GPU exchange: out_keys [nq][k] uint64,
 * key = orderable(score) << 32 | ~(row id + id_offset) — larger key = better (higher score,
 * then lower global id); 0 = no entry.  One 8-byte word per result is what the ranks
 * all-gather (dense_retriever.py:48-69 exchanges through the file system instead).
 * id_offset + rows must stay below 2^32 - 1. */
int vr_index_search_keys(vr_index_t ix, const float* queries, int32_t nq, int32_t k, int64_t id_offset,
                         uint64_t* out_keys, int32_t on_device, void* stream);
/* Merge per-shard results (e.g. after an RCCL all-gather): in [n_parts][nq][k] scores and
 * global ids -> out [nq][k], same ordering rule.  Device pointers. */
int vr_topk_merge(int device_id, const float* scores, const int64_t* ids, int32_t n_parts,
                  int32_t nq, int32_t k, float* out_scores, int64_t* out_ids, void* stream);
/* The same merge over the packed keys of vr_index_search_keys ([n_parts][nq][k], e.g. the
 * all-gather's output buffer as it is) -> scores and global ids. */
int vr_topk_merge_keys(int device_id, const uint64_t* keys, int32_t n_parts, int32_t nq, int32_t k,
                       float* out_scores, int64_t* out_ids, void* stream);

/* ---- synthetic corpus (bench / test support; no reference counterpart) --------------------- */
/* Pages first .. first + n - 1 of the deterministic synthetic corpus of visrag_amd/synth.py::synth_pages,
 * bit-identical to the host generator, written as uint8 [n][size][size][3] to DEVICE memory.  BASELINE
 * config 3 needs 100 000 distinct pages; there is no dataset on the box (BASELINE.json: "data": synthetic). */
int vr_synth_pages(int device_id, uint8_t* out, int32_t n, int32_t size, int64_t seed, int64_t first, void* stream);

/* ---- streams (caller support; no reference counterpart) ------------------------------------ */
/* *overlap = 1 when kernels on the two streams run SIDE BY SIDE, 0 when the runtime put the streams on one hardware
 * queue (kernels of one wait for the other's).  Probed with two 400 us one-thread kernels, three rounds (~2.5 ms);
 * both streams are synchronised.  A caller that keeps two vr_model_clone workspaces in flight picks its stream pair
 * with this (visrag_amd/engine.py::overlapping_streams): a pair on one queue runs at the single-stream rate. */
int vr_streams_overlap(int device_id, void* stream_a, void* stream_b, int32_t* overlap);

/* ---- host pre-processing moved to the GPU ------------------- */
/* Bicubic resize of an 8-bit RGB (HWC) image, bit-exact with Pillow's
 * Image.resize((out_w, out_h), Image.Resampling.BICUBIC) — the resize of slice_image /
 * find_best_resize (modeling_minicpmv.py:482-537).  src on host or device, dst on device. */
int vr_resize_bicubic(int device_id, const uint8_t* src, int32_t src_on_device, int32_t H, int32_t W,
                      uint8_t* dst, int32_t out_h, int32_t out_w, void* stream);

/* ---- single kernels (parity tests and micro-benchmarks call these through the ABI) ---- */
/* out[M][N] = epilogue(A[M][K] * W[N][K]^T).  A, W bf16 row-major, K % 64 == 0,
 * N % 128 == 0, buffers padded to a multiple of 128 rows.  epilogue:
 *   0 bf16 out = acc + bias            3 f32 out = resid + alpha*(acc + bias)
 *   1 bf16 out = gelu_erf(acc + bias)  4 bf16 out[N/2] = silu(gate)*up (16-row interleaved W)
 *   2 f32  out = acc + bias            5 bf16 out = rope(acc) for col < rope_cols (head 64)
 * bias (f32[N]) and resid (f32[M][ldo]) may be NULL.  variant: 3 = the engine's own choice,
 * 0 = 128x128 tile, 7 = 256x192 tile (N % 192 == 0), 9 = 256x256 tile (8 waves), 12 = 256x256 tile with one
 * wave per SIMD (what the engine uses for its big GEMMs), 13 = its 256x192 form (N % 192 == 0; epilogues 2, 3).
 * Buffers padded to a multiple of 256 rows for the 256-row tiles. */
int vr_op_gemm(int device_id, const void* A, int32_t lda, const void* W, int32_t ldw,
               int32_t M, int32_t N, int32_t K, int32_t epilogue, const float* bias,
               const float* resid, float alpha, void* out, int32_t ldo,
               const int32_t* rope_pos, const float* rope_table, int32_t rope_cols,
               int32_t variant, void* stream);
/* y = LN(x) (kind 0, affine, eps) or RMSNorm(x) (kind 1): x f32 [rows][dim] -> bf16 [rows][ldo]. */
int vr_op_norm(int device_id, int32_t kind, const float* x, int32_t rows, int32_t dim,
               const float* weight, const float* bias, float eps, void* out, int32_t ldo,
               void* stream);
/* Flash attention over bf16 q/k/v with row strides ld*, per-batch row ranges cu_q/cu_kv
 * ([B+1], device), head_dim in {64,72,128}; causal uses absolute positions within the
 * sequence.  q_batch_stride==0 shares q across the batch (resampler). out bf16 [rows_q][ldo]. */
int vr_op_attention(int device_id, const void* q, int32_t ldq, const void* k, int32_t ldk,
                    const void* v, int32_t ldv, void* out, int32_t ldo,
                    const int32_t* cu_q, const int32_t* cu_kv, int32_t B, int32_t heads,
                    int32_t head_dim, int32_t max_q, int32_t causal, int32_t q_shared,
                    float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VISRAG_HIP_H */
