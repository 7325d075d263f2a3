// The EVisRAG generator's model object, shared by gen.hip (language model) and gen_vision.hip (vision tower).
#pragma once
#include <string>
#include <vector>

#include "../../include/visrag_gen.h"
#include "engine_common.h"

struct GenLayer {
    Vec ln1, ln2;
    Linear qkv, o, gu, down;
    int parts_w = 0, parts_b = 0, parts_gu = 0;
};

constexpr int GEN_KS_MAX = 64;      // split-K planes of the decode path
constexpr int GEN_RUN_RING = 8;     // most free-running steps in flight before their tokens are collected

struct VisionTower;                 // gen_vision.hip

struct vg_model_s {
    int device = 0;
    vg_config_t c{};
    bool finalized = false;
    int E = 0, H = 0, KV = 0, I = 0, V = 0, QKV = 0, QD = 0, KVD = 0;
    DevBuf embed;
    bool has_embed = false;
    std::vector<GenLayer> layers;
    Vec final_norm;
    Linear lm_head;
    DevBuf inv_freq;
    std::vector<DevBuf> kc, vc;          // per layer [n_slots * max_len][KVD] bf16: slot s owns rows [s * max_len, (s + 1) * max_len)
    int n_slots = 1, cur = 0;            // sequences the model holds; the one the single-sequence entry points work on
    std::vector<int> slot_len;           // cache rows in use per slot (the current slot's lives in `len`)
    std::vector<char> slot_logits;       // per slot: logits are on the device
    int len = 0;                         // rows of the CURRENT slot's cache in use
    bool have_logits = false;            // the current slot has logits on the device
    DevBuf w_batch, w_logits_b;          // vg_decode_batch: the step's device-side tables; logits rows of the batch
    int Tcap = 0;
    DevBuf w_h, w_xn, w_qkv, w_q, w_att, w_act, w_last, w_part, w_logits, w_ids, w_pos, w_cu, w_seen, w_tok, w_erows, w_emb;
    DevBuf w_attp, w_lse;               // decode: partial attention rows [GEN_ATT_SPLITS][QD] bf16 + their log-sum-exps
    DevBuf w_state;                     // GenState: token / positions / cache length / KV ranges of the decode step, on the device
    bool tok_on_device = false;         // w_state.token holds the last sampled token (vg_sample / a free-running step)
    // free-running generation (vg_run_*): one decode + sample step captured as a hipGraph, replayed per token
    hipStream_t run_stream = nullptr;   // capture stream; also the stream of a run when the caller passes none
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    float g_temp = -1.f, g_pen = -1.f;  // sampling parameters baked into the captured step
    unsigned long long g_seed = 0;
    bool running = false;
    hipStream_t run_on = nullptr;       // stream of the current run
    int run_steps = 0;                  // steps enqueued in the current run
    int* h_tokens = nullptr;            // pinned ring [GEN_RUN_RING] the sampled tokens are copied into
    hipEvent_t run_ev[8] = {};          // run_ev[i % 8]: step i's token has landed
    // the decode step's persistent layer kernel (gen_persist.hip): layer table for the CURRENT slot, barrier words, host-mapped abort flag
    DevBuf p_table, p_sync, p_ss;
    unsigned* p_abort = nullptr;        // hipHostMalloc'd
    int p_grid = 0;                     // 0: the shape / device does not take the kernel (separate launches instead)
    VisionTower* vis = nullptr;         // attached by vg_vision_create
    int vis_tokens = 0;                 // embedding rows the last vg_vision_encode left in w_emb (image-token order)
};

// per-slot views of the sequence state
static inline void* gen_kc(vg_model_s* m, int l, int slot) { return (char*)m->kc[l].p + (size_t)slot * m->c.max_len * m->KVD * 2; }
static inline void* gen_vc(vg_model_s* m, int l, int slot) { return (char*)m->vc[l].p + (size_t)slot * m->c.max_len * m->KVD * 2; }
static inline float* gen_logits(vg_model_s* m, int slot) { return m->w_logits.as<float>() + (size_t)slot * m->V; }
static inline unsigned* gen_seen(vg_model_s* m, int slot) { return m->w_seen.as<unsigned>() + (size_t)slot * ((m->V + 31) / 32); }

static inline GemmArgs gen_gemm_args(const void* A, int lda, const Linear& L, int M, void* out, int ldo) {
    GemmArgs a{};
    a.A = A; a.lda = lda; a.W = L.w.p; a.ldw = L.k_pad; a.M = M; a.N = L.n_pad; a.K = L.k_pad;
    a.bias = L.has_b ? L.b.as<float>() : nullptr;
    a.out = out; a.ldo = ldo; a.alpha = 1.0f;
    return a;
}

// gen_vision.hip: the tower's side of vg_load_weight / vg_finalize / vg_destroy
int vision_load_weight(vg_model_s* m, const std::string& key, const void* dev_src, int is_bf16, const int64_t* shape, int ndim,
                       size_t numel);                 // key without the "model.visual." prefix
int vision_check_complete(const vg_model_s* m);
void vision_destroy(vg_model_s* m);
