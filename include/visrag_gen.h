/* C ABI of the EVisRAG generator's LANGUAGE MODEL in libvisrag_hip.so (BASELINE config 5).
 *
 * What it stands in for: the reference hands generation to vLLM —
 *   src/evisrag/predict.py:112-117   llm = LLM(model=..., tensor_parallel_size=1, dtype="bfloat16",
 *                                              limit_mm_per_prompt={"image": 5, "video": 0})
 *   src/evisrag/predict.py:119-123   SamplingParams(temperature=..., repetition_penalty=1.05, max_tokens=2048)
 *   src/evisrag/predict.py:147       llm.generate(batch_input, sampling_params)      (one prompt at a time, :128)
 * vLLM is not vendored (vllm==0.9.1, EVisRAG_requirements.txt:236); the model is Qwen2.5-VL (EVisRAG-7B).  This ABI
 * covers the decoder — prefill over the prompt's token embeddings, KV cache, one-token decode steps, vLLM's logits
 * processing (repetition penalty over prompt + output, temperature; 0 = greedy) — and the vision tower that turns the
 * page images of predict.py:98-103,140 into the embedding rows of the prompt's image tokens (vg_vision_*).
 * visrag_amd/evisrag.py mirrors the LLM / SamplingParams / generate call sites on top of it.
 *
 * Same conventions as visrag_hip.h: plain pointers and sizes, 0 = OK, vr_last_error() for the message.
 * State-dict keys of Qwen2_5_VLForConditionalGeneration ("model.language_model.*", "lm_head.weight", and with a tower
 * attached "model.visual.*") are consumed verbatim by vg_load_weight. */
#ifndef VISRAG_GEN_H
#define VISRAG_GEN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vg_model_s* vg_model_t;

typedef struct {
    int32_t hidden_size;            /* 3584 */
    int32_t num_layers;             /* 28 */
    int32_t num_heads;              /* 28 query heads, head_dim = hidden / heads must be 128 */
    int32_t num_kv_heads;           /* 4 */
    int32_t intermediate_size;      /* 18944 */
    int32_t vocab_size;             /* 152064 */
    int32_t max_len;                /* KV-cache rows (prompt + generated tokens) */
    int32_t max_prefill;            /* most prompt tokens one vg_prefill call may carry */
    float rms_norm_eps;             /* 1e-6 */
    float rope_theta;               /* 1e6 */
    int32_t mrope_section[3];       /* rotary channel pairs that take the temporal / height / width position: 16, 24, 24 */
    int32_t max_seqs;               /* sequences the model can hold at once (KV cache, logits, seen-token set per slot);
                                     * 0 or 1 = one sequence, at most 16 (vg_select / vg_decode_batch) */
} vg_config_t;

int vg_create(int device_id, const vg_config_t* cfg, vg_model_t* out);
int vg_destroy(vg_model_t m);
/* The vision tower (Qwen2_5_VisionTransformerPretrainedModel; HF modeling_qwen2_5_vl.py:345-470). */
typedef struct {
    int32_t depth;                  /* 32 */
    int32_t hidden_size;            /* 1280 */
    int32_t num_heads;              /* 16: head_dim 80 (a multiple of 4, at most 128) */
    int32_t intermediate_size;      /* 3420 */
    int32_t out_hidden_size;        /* 3584 = the language model's hidden_size */
    int32_t in_channels;            /* 3 */
    int32_t patch_size;             /* 14 */
    int32_t temporal_patch_size;    /* 2 */
    int32_t spatial_merge_size;     /* 2: 2 x 2 patches become one image token */
    int32_t window_size;            /* 112 pixels */
    int32_t n_fullatt;              /* 4 */
    int32_t fullatt_blocks[16];     /* 7, 15, 23, 31: the blocks that attend over whole images */
    int32_t max_rows;               /* most patch rows (sum of t*h*w over the images) of one vg_vision_encode call */
    float rms_norm_eps;             /* 1e-6 */
} vg_vision_config_t;

/* Attach a tower to a model (before its weights are loaded; without one, "model.visual.*" tensors are skipped). */
int vg_vision_create(vg_model_t m, const vg_vision_config_t* cfg);
/* Run the tower.
 *   pixels     host f32 [rows][in_channels * temporal_patch_size * patch_size^2]: the image processor's pixel_values
 *              (rows in its merge-block-major order, all images concatenated; converted to bf16 on the device like the
 *              reference's bf16 model does)
 *   grid_thw   host int32 [n_images][3]: frames, patch rows, patch columns of every image (image_grid_thw)
 *   embeds_out host f32 [rows / merge^2][out_hidden_size] or NULL: the image tokens' embeddings, image by image
 * The embeddings also stay on the device: a vg_prefill with embeds == NULL and n_embed == rows / merge^2 uses them. */
int vg_vision_encode(vg_model_t m, const float* pixels, const int32_t* grid_thw, int32_t n_images, float* embeds_out,
                     void* stream);
/* The same with the image processor's rescale / normalise / patchify done on the device: pages[i] is an 8-bit RGB (HWC)
 * image of exactly (grid_h * patch_size) x (grid_w * patch_size) pixels — the page already resized to smart_resize's size,
 * e.g. by vr_resize_bicubic (Pillow-exact) — all on the host or all on the device; mean3 / std3 are the processor's
 * image_mean / image_std (host).  Element (c, t, y, x) of a patch row = (u8 / 255 - mean[c]) / std[c] in fp32, rounded to
 * bf16 — what the reference's processor + bf16 tower input amount to (predict.py:140; 0.6 MB per 448 x 448 page over the
 * boundary instead of 4.8 MB of f32 rows).  Still images only (t = 1). */
int vg_vision_encode_pages(vg_model_t m, const uint8_t* const* pages, int32_t pages_on_device, const float* mean3,
                           const float* std3, const int32_t* grid_thw, int32_t n_images, float* embeds_out, void* stream);
/* Host-only (no GPU needed): the tower's token geometry for a set of grids — tests.  order [tokens]: the image token at
 * window-order place i; win_bounds [n_windows + 1]: patch-row boundaries of the attention windows in that order;
 * hw [rows][2]: patch coordinates of every pixel row.  Any output may be NULL. */
int vg_vision_plan(const vg_vision_config_t* cfg, const int32_t* grid_thw, int32_t n_images, int32_t* order,
                   int32_t* win_bounds, int32_t* n_windows, int32_t* hw);

/* one tensor of the HF state dict ("model.language_model.layers.0.self_attn.q_proj.weight", "lm_head.weight", ...);
 * dtype / on_device as in vr_model_load_weight (0 = f32, 1 = bf16; data on the host or on this device) */
int vg_load_weight(vg_model_t m, const char* name, const void* data, const int64_t* shape, int32_t ndim, int32_t dtype,
                   int32_t on_device);
int vg_finalize(vg_model_t m);      /* checks that every tensor arrived */

/* Start a sequence: empties the KV cache and the seen-token set, then runs the prompt.
 *   ids          [T] host int32: token ids (their embedding rows are gathered; all of them count as "seen" for the
 *                repetition penalty, placeholders included — like the prompt_token_ids vLLM penalises)
 *   embed_rows   [n_embed] host int32 (or NULL): prompt positions whose embedding is replaced ...
 *   embeds       ... by row i of this host f32 [n_embed][hidden] matrix (the image tokens); NULL = the rows the last
 *                vg_vision_encode left on the device (n_embed must equal their number)
 *   pos3         [3][T] host int32: temporal / height / width position of every token (get_rope_index)
 * Leaves the last token's logits on the device for vg_sample. */
int vg_prefill(vg_model_t m, const int32_t* ids, int32_t T, const int32_t* embed_rows, const float* embeds, int32_t n_embed,
               const int32_t* pos3, void* stream);
/* Next token from the current logits: repetition penalty over every seen id, then temperature sampling (temperature
 * 0 = argmax).  The token is marked seen; `step` (the index of the generated token) and `seed` select the noise. */
int vg_sample(vg_model_t m, float temperature, float repetition_penalty, uint64_t seed, int32_t step, int32_t* token_out,
              void* stream);
/* Append one token at position pos[3] (all three equal for generated text) and compute its logits. */
int vg_decode(vg_model_t m, int32_t token, const int32_t pos[3], void* stream);
/* Free-running generation: the steps after the first token without a host round trip per token.  After vg_prefill +
 * vg_sample (whose token stays on the device), vg_run_begin fixes the sampling parameters and the position of the first
 * appended token; every vg_run_step ENQUEUES one step — append the last sampled token at the next position, compute
 * its logits, sample the next token with the rules of vg_sample (noise index first_step, first_step + 1, ...) — as one
 * captured hipGraph launch and returns at once; vg_run_token(i) waits for step i's token (steps are numbered from 0
 * within the run; at most 8 may be in flight uncollected).  Token, positions and cache length advance on the device.
 * A caller that stops on an end-of-sequence token simply stops enqueuing (a step already in flight appends one unused
 * row).  vg_run_end — or any other vg_* call on the model — waits for the steps in flight.
 * stream NULL = a stream the model owns. */
int vg_run_begin(vg_model_t m, int32_t position, float temperature, float repetition_penalty, uint64_t seed,
                 int32_t first_step, void* stream);
int vg_run_step(vg_model_t m);
int vg_run_token(vg_model_t m, int32_t index, int32_t* token);
int vg_run_end(vg_model_t m);
/* Several sequences in one model (max_seqs slots, each with its own KV cache, logits and seen-token set) — what
 * vLLM does with the list `llm.generate` is given (predict.py:147 passes one prompt at a time; a caller that has several
 * queries pending can pass them together).  vg_select makes `slot` the sequence that vg_prefill / vg_vision_encode +
 * vg_prefill / vg_sample / vg_decode / vg_run_* / vg_logits / vg_cache_len work on (slot 0 after vg_create).
 * vg_decode_batch appends ONE token to each of n distinct slots (every one with a sequence in progress) as a single
 * step: every weight matrix is streamed once for the n rows (the decode step is bound by that stream), attention runs
 * per sequence over its own cache.  pos = [n][3] positions.  vg_sample_batch samples the next token of each of the n
 * slots from its logits with the rules of vg_sample.  Rows are computed independently of each other: a sequence's
 * tokens do not depend on what it is batched with. */
int vg_select(vg_model_t m, int32_t slot);
int vg_decode_batch(vg_model_t m, int32_t n, const int32_t* slots, const int32_t* tokens, const int32_t* pos, void* stream);
int vg_sample_batch(vg_model_t m, int32_t n, const int32_t* slots, float temperature, float repetition_penalty, uint64_t seed,
                    int32_t step, int32_t* tokens_out, void* stream);
/* copy the current logits (f32 [vocab]) to the host — tests */
int vg_logits(vg_model_t m, float* out, void* stream);
int vg_cache_len(vg_model_t m, int32_t* len);

#ifdef __cplusplus
}
#endif
#endif
