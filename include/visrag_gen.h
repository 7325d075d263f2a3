/* C ABI of the EVisRAG generator's LANGUAGE MODEL in libvisrag_hip.so (BASELINE config 5).
 *
 * What it stands in for: the reference hands generation to vLLM —
 *   src/evisrag/predict.py:112-117   llm = LLM(model=..., tensor_parallel_size=1, dtype="bfloat16",
 *                                              limit_mm_per_prompt={"image": 5, "video": 0})
 *   src/evisrag/predict.py:119-123   SamplingParams(temperature=..., repetition_penalty=1.05, max_tokens=2048)
 *   src/evisrag/predict.py:147       llm.generate(batch_input, sampling_params)      (one prompt at a time, :128)
 * vLLM is not vendored (vllm==0.9.1, EVisRAG_requirements.txt:236); the model is Qwen2.5-VL (EVisRAG-7B).  This ABI
 * covers the decoder: prefill over the prompt's token embeddings (image tokens arrive as embedding rows — the vision
 * tower is not part of this library yet), KV cache, one-token decode steps, and vLLM's logits processing
 * (repetition penalty over prompt + output, temperature; 0 = greedy).  visrag_amd/evisrag.py mirrors the
 * LLM / SamplingParams / generate call sites on top of it.
 *
 * Same conventions as visrag_hip.h: plain pointers and sizes, 0 = OK, vr_last_error() for the message.
 * State-dict keys of Qwen2_5_VLForConditionalGeneration's language model are consumed verbatim by vg_load_weight. */
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
} vg_config_t;

int vg_create(int device_id, const vg_config_t* cfg, vg_model_t* out);
int vg_destroy(vg_model_t m);
/* one tensor of the HF state dict ("model.language_model.layers.0.self_attn.q_proj.weight", "lm_head.weight", ...);
 * dtype / on_device as in vr_model_load_weight (0 = f32, 1 = bf16; data on the host or on this device) */
int vg_load_weight(vg_model_t m, const char* name, const void* data, const int64_t* shape, int32_t ndim, int32_t dtype,
                   int32_t on_device);
int vg_finalize(vg_model_t m);      /* checks that every tensor arrived */

/* Start a sequence: empties the KV cache and the seen-token set, then runs the prompt.
 *   ids          [T] host int32: token ids (their embedding rows are gathered; all of them count as "seen" for the
 *                repetition penalty, placeholders included — like the prompt_token_ids vLLM penalises)
 *   embed_rows   [n_embed] host int32 (or NULL): prompt positions whose embedding is replaced ...
 *   embeds       ... by row i of this host f32 [n_embed][hidden] matrix (the image tokens)
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
/* copy the current logits (f32 [vocab]) to the host — tests */
int vg_logits(vg_model_t m, float* out, void* stream);
int vg_cache_len(vg_model_t m, int32_t* len);

#ifdef __cplusplus
}
#endif
#endif
