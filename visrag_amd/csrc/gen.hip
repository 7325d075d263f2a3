// Host side of the EVisRAG generator's language model in libvisrag_hip.so: the C ABI of include/visrag_gen.h.
// Reference boundary: src/evisrag/predict.py:112-123,147 (vllm.LLM / SamplingParams / llm.generate; vLLM itself is not
// vendored).  Architecture: the Qwen2.5-VL text decoder — per layer RMSNorm, q/k/v projections with bias, multimodal
// RoPE, grouped-query causal attention over a KV cache, o projection, RMSNorm, SwiGLU MLP; final RMSNorm + lm_head
// (HF modeling_qwen2_5_vl.py:602-758; restated on the CPU in oracle/qwen_gen_oracle.py).
//
// Two paths through the same kernels of this library:
//   prefill (M = prompt tokens): the one-wave-per-SIMD GEMMs with bias / residual / SwiGLU epilogues, fp32 residual
//     stream, flash attention with a KV-group stride reading K / V straight from the cache;
//   decode (M = 1): every GEMM is one pass over its weights (gemm_skinny.hip: 256 columns x a K range per workgroup,
//     four LDS stages of LDS-DMA, eight MFMAs per wave and K-step), split over K so that ~500 workgroups pull on the
//     HBM; the fp32 partial rows are summed by the kernel that consumes them (mRoPE for q/k/v, SwiGLU for gate/up,
//     the residual-adding RMSNorm for o / down).  ~14 GB of bf16 weights per token at 7B: the HBM sets the floor.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <string>
#include <vector>

#include "gen_model.h"

static int end_run(vg_model_s* m);      // waits for the steps of a free run (vg_run_*) still in flight

// the captured decode step holds device pointers by value: anything that may move a buffer drops it
static void drop_graph(vg_model_s* m) {
    if (m->graph_exec) { (void)hipGraphExecDestroy(m->graph_exec); m->graph_exec = nullptr; }
    if (m->graph) { (void)hipGraphDestroy(m->graph); m->graph = nullptr; }
}

// decode (gemm_skinny.hip): how many K ranges a GEMM's 256-column tiles are cut into.  One workgroup fits on a CU (its
// four LDS stages), so the fastest shape is ONE balanced round of workgroups with K loops as long as that allows —
// measured on the 7B layer (ms per token): gate/up 148 tiles as 148 x 56 K-steps 3.74, 296 x 28 (a second round of 40)
// 3.94, 592 x 14 3.80; down 14 tiles as 252 x 17 3.74, 112 x 37 3.94, 518 x 8 4.08.  The splits need not divide the
// K-steps (the last range is shorter); every range keeps >= 4 steps; more tiles than CUs: no split.
#ifndef VR_KS_QKV
#define VR_KS_QKV 0
#define VR_KS_O 0
#define VR_KS_GU 0
#define VR_KS_DOWN 0
#endif
static int choose_ksplit(int n, int k, int forced = 0) {
    if (forced > 0) return forced;
    const int tiles = (n + 255) / 256, nk = k / 64;
    int best = 1;
    for (int d = 2; d <= GEN_KS_MAX && tiles * d <= 256; ++d) {
        const int per = (nk + d - 1) / d;
        if (per >= 4 && (d - 1) * per < nk) best = d;
    }
    return best;
}

// The persistent layer kernel of the decode step (gen_persist.hip): decide whether this model / device takes it and build
// its layer table for the CURRENT slot (the caches are per slot).  It is OPT-IN — VR_DECODE_PERSIST=1 in the environment
// when the model is finalized / a slot is selected (or -DVR_DECODE_PERSIST=1) — because on MI355X it only draws level with
// the separate launches (3.33 against 3.29 ms per token at the 7B shape: its weight streams run at 6.5 TB/s instead of
// 5.4-5.7, but seven grid barriers per layer cost 4.3 us each under the prefetch traffic).
#ifndef VR_DECODE_PERSIST
#define VR_DECODE_PERSIST 0
#endif
static int persist_setup(vg_model_s* m) {
    m->p_grid = 0;
    const char* env = getenv("VR_DECODE_PERSIST");
    if (env ? atoi(env) == 0 : !VR_DECODE_PERSIST) return VR_OK;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, m->device) != hipSuccess || decode_persist_occupancy() < 1) return VR_OK;
    const int grid = prop.multiProcessorCount;
    const int E = m->E, QD = m->QD, QKV = m->QKV, G = m->H / m->KV;
    const GenLayer& L0 = m->layers[0];
    const int ks_qkv = choose_ksplit(QKV, E, VR_KS_QKV), ks_o = choose_ksplit(E, QD, VR_KS_O);
    const int ks_d = choose_ksplit(E, L0.down.k_pad, VR_KS_DOWN), ks_gu = choose_ksplit(L0.gu.n_pad, E, VR_KS_GU);
    auto per64 = [](int k, int ks) { return ((k / 64 + ks - 1) / ks) * 64; };
    const bool fits =
        E <= 4096 && (E / 4 + 63) / 64 <= 16 && (E / 4 + 63) / 64 <= grid && G <= 16 && GEN_ATT_SPLITS * m->KV <= grid && ks_gu == 1 &&
        L0.qkv.n_pad == QKV && L0.qkv.k_pad == E && L0.o.n_pad == E && L0.o.k_pad == QD && L0.gu.k_pad == E && L0.down.n_pad == E &&
        L0.down.k_pad == pad128(m->I) && !L0.o.has_b && !L0.gu.has_b && !L0.down.has_b &&
        (QKV + 255) / 256 * ks_qkv <= grid && (E + 255) / 256 * ks_o <= grid && (E + 255) / 256 * ks_d <= grid && (L0.gu.n_pad + 255) / 256 <= grid &&
        per64(E, ks_qkv) <= 4096 && per64(QD, ks_o) <= 4096 && per64(L0.down.k_pad, ks_d) <= 4096 &&
        (size_t)std::max(ks_qkv, std::max(ks_o, ks_d)) * std::max(QKV, E) * 4 <= m->w_part.bytes;
    if (!fits) return VR_OK;
    std::vector<PersistLayer> t(m->layers.size());
    for (size_t l = 0; l < t.size(); ++l) {
        GenLayer& L = m->layers[l];
        t[l] = PersistLayer{L.qkv.w.p, L.o.w.p, L.gu.w.p, L.down.w.p, L.qkv.has_b ? L.qkv.b.as<float>() : nullptr,
                            L.ln1.v.as<float>(), L.ln2.v.as<float>(), gen_kc(m, (int)l, m->cur), gen_vc(m, (int)l, m->cur)};
    }
    if (!m->p_table.p) {
        VRCHK(m->p_table.alloc(t.size() * sizeof(PersistLayer)));
        VRCHK(m->p_sync.alloc(128 * 8));
        VRCHK(m->p_ss.alloc(16 * 4));
        HIPCHK(hipMemset(m->p_sync.p, 0, 128 * 8));
        HIPCHK(hipMemset(m->p_ss.p, 0, 16 * 4));
        HIPCHK(hipHostMalloc((void**)&m->p_abort, 64, hipHostMallocMapped));
        *m->p_abort = 0;
    }
    HIPCHK(hipMemcpy(m->p_table.p, t.data(), t.size() * sizeof(PersistLayer), hipMemcpyHostToDevice));
    m->p_grid = grid;
    return VR_OK;
}
static int persist_layers(vg_model_s* m, hipStream_t s) {
    const GenLayer& L0 = m->layers[0];
    PersistArgs a{};
    a.layers = m->p_table.as<PersistLayer>(); a.n_layers = (int)m->layers.size();
    a.E = m->E; a.QKV = m->QKV; a.QD = m->QD; a.KVD = m->KVD; a.H = m->H; a.KV = m->KV; a.Ip = L0.down.k_pad; a.N2 = L0.gu.n_pad;
    a.ldw_qkv = L0.qkv.k_pad; a.ldw_o = L0.o.k_pad; a.ldw_gu = L0.gu.k_pad; a.ldw_d = L0.down.k_pad;
    a.ks_qkv = choose_ksplit(m->QKV, m->E, VR_KS_QKV); a.ks_o = choose_ksplit(m->E, m->QD, VR_KS_O);
    a.ks_d = choose_ksplit(m->E, L0.down.k_pad, VR_KS_DOWN);
    a.eps = m->c.rms_norm_eps; a.g_final = m->final_norm.v.as<float>();
    a.h = m->w_h.as<float>(); a.xn = m->w_xn.p; a.planes = m->w_part.as<float>(); a.act = m->w_act.p;
    a.attp = m->w_attp.p; a.lse = m->w_lse.as<float>(); a.ss = m->p_ss.as<float>();
    a.st = m->w_state.as<GenState>(); a.inv_freq = m->inv_freq.as<float>();
    a.sec_t = m->c.mrope_section[0]; a.sec_h = m->c.mrope_section[1];
    a.sync = m->p_sync.as<unsigned long long>(); a.abort_host = m->p_abort;
    HIPCHK(launch_decode_persist(a, m->p_grid, s));
    return VR_OK;
}
// a barrier of the persistent kernel timed out (the workgroups were not all resident): everything after it is garbage
// The kernel's abort flag and its barrier counters are sticky: leave the persistent path for good — the step that timed out
// is reported as failed (its logits are garbage), every later step runs as separate launches (p_grid = 0) on clean state.
static int persist_check(vg_model_s* m) {
    if (m->p_abort && *m->p_abort) {
        *m->p_abort = 0;
        m->p_grid = 0;
        if (m->p_sync.p) (void)hipMemset(m->p_sync.p, 0, 128 * 8);
        if (m->p_ss.p) (void)hipMemset(m->p_ss.p, 0, 16 * 4);
        return fail(VR_ERR_HIP, "the decode kernel's grid barrier timed out (its workgroups must all be resident: one per CU — "
                                "another kernel was running on the device?); this step's result is invalid, the model has fallen "
                                "back to separate launches for the steps that follow: prefill again and continue");
    }
    return VR_OK;
}

extern "C" int vg_create(int device_id, const vg_config_t* cfg, vg_model_t* out) {
    if (!cfg || !out) return fail(VR_ERR_INVALID, "cfg/out is NULL");
    const vg_config_t& c = *cfg;
    if (c.num_heads <= 0 || c.hidden_size != c.num_heads * 128) return fail(VR_ERR_INVALID, "head_dim must be 128 (hidden %d, heads %d)", c.hidden_size, c.num_heads);
    if (c.num_kv_heads <= 0 || c.num_heads % c.num_kv_heads) return fail(VR_ERR_INVALID, "num_heads must be a multiple of num_kv_heads");
    if (c.hidden_size % 256 || c.intermediate_size % 64 || c.vocab_size % 128) return fail(VR_ERR_INVALID, "hidden %% 256, intermediate %% 64, vocab %% 128 must be 0");
    // the row-norm kernels hold a row in 14 float4 registers per lane (norm.hip: NORM_MAXV): 3584 columns — the 3B and 7B
    // models; the 32B / 72B widths (5120 / 8192) would fail inside the first prefill instead
    if (c.hidden_size > 3584) return fail(VR_ERR_INVALID, "hidden_size %d is beyond this build's row kernels (<= 3584: Qwen2.5-VL-3B / 7B)", c.hidden_size);
    if (c.mrope_section[0] + c.mrope_section[1] + c.mrope_section[2] != 64) return fail(VR_ERR_INVALID, "mrope sections must add up to 64 channel pairs");
    if (c.num_layers <= 0 || c.max_len <= 0 || c.max_prefill <= 0 || c.max_prefill > c.max_len) return fail(VR_ERR_INVALID, "bad layer / length settings");
    if (c.max_seqs < 0 || c.max_seqs > 16) return fail(VR_ERR_INVALID, "max_seqs %d: 0 / 1 (one sequence) up to 16", c.max_seqs);
    VRCHK(set_dev(device_id));
    vg_model_s* m = new vg_model_s();
    m->device = device_id;
    m->c = c;
    m->E = c.hidden_size; m->H = c.num_heads; m->KV = c.num_kv_heads; m->I = c.intermediate_size; m->V = c.vocab_size;
    m->QD = m->H * 128; m->KVD = m->KV * 128; m->QKV = m->QD + 2 * m->KVD;
    m->layers.resize(c.num_layers);
    m->kc.resize(c.num_layers);
    m->vc.resize(c.num_layers);
    m->n_slots = std::max(1, c.max_seqs);
    m->slot_len.assign(m->n_slots, 0);
    m->slot_logits.assign(m->n_slots, 0);
    *out = m;
    auto bail = [&](int rc) { vg_destroy(m); *out = nullptr; return rc; };
    // rotary frequencies exactly as the reference builds them: 1 / theta^(2p / 128) in fp32 (modeling_qwen2_5_vl.py:519-523)
    std::vector<float> inv(64);
    for (int p = 0; p < 64; ++p) inv[p] = 1.0f / powf(c.rope_theta, (float)(2 * p) / 128.0f);
    int rc;
    if ((rc = m->inv_freq.alloc(64 * 4)) != VR_OK) return bail(rc);
    if (hipMemcpy(m->inv_freq.p, inv.data(), 64 * 4, hipMemcpyHostToDevice) != hipSuccess) return bail(fail(VR_ERR_HIP, "hipMemcpy failed"));
    for (int l = 0; l < c.num_layers; ++l) {
        if ((rc = m->kc[l].alloc((size_t)m->n_slots * c.max_len * m->KVD * 2)) != VR_OK) return bail(rc);
        if ((rc = m->vc[l].alloc((size_t)m->n_slots * c.max_len * m->KVD * 2)) != VR_OK) return bail(rc);
    }
    const size_t T = (size_t)pad256(c.max_prefill);
    m->Tcap = (int)T;
    const size_t E = m->E;
    struct { DevBuf* b; size_t bytes; } ws[] = {
        {&m->w_h, T * E * 4}, {&m->w_xn, T * E * 2}, {&m->w_qkv, T * m->QKV * 2}, {&m->w_q, T * m->QD * 2},
        {&m->w_att, T * m->QD * 2}, {&m->w_act, T * (size_t)pad128(m->I) * 2}, {&m->w_last, 256 * E * 2},
        {&m->w_part, (size_t)GEN_KS_MAX * std::max<size_t>(std::max<size_t>(m->QKV, E), (size_t)pad128(2 * m->I)) * 4}, {&m->w_logits, (size_t)m->n_slots * m->V * 4},
        {&m->w_ids, T * 4}, {&m->w_pos, 3 * T * 4}, {&m->w_cu, 4 * 4},
        {&m->w_attp, (size_t)m->n_slots * GEN_ATT_SPLITS * m->QD * 2}, {&m->w_lse, (size_t)m->n_slots * GEN_ATT_SPLITS * m->H * 4}, {&m->w_seen, (size_t)m->n_slots * ((m->V + 31) / 32) * 4},
        {&m->w_tok, 16 + 64 * 8}, {&m->w_erows, T * 4}, {&m->w_emb, T * E * 4},
        {&m->w_state, sizeof(GenState)}, {&m->w_batch, sizeof(GenBatch)}, {&m->w_logits_b, (size_t)(m->n_slots > 1 ? m->n_slots : 0) * m->V * 4}};
    for (auto& w : ws)
        if ((rc = w.b->alloc(w.bytes)) != VR_OK) return bail(rc);
    return VR_OK;
}

extern "C" int vg_destroy(vg_model_t m) {
    if (!m) return VR_OK;
    (void)hipSetDevice(m->device);
    if (m->run_stream) (void)hipStreamSynchronize(m->run_stream);
    drop_graph(m);
    for (auto& e : m->run_ev)
        if (e) (void)hipEventDestroy(e);
    if (m->h_tokens) (void)hipHostFree(m->h_tokens);
    if (m->p_abort) (void)hipHostFree(m->p_abort);
    vision_destroy(m);
    delete m;                           // DevBuf destructors release everything
    return VR_OK;
}

extern "C" int vg_load_weight(vg_model_t m, const char* name_c, const void* data, const int64_t* shape, int32_t ndim,
                              int32_t dtype, int32_t on_device) {
    if (!m || !name_c || !data || !shape) return fail(VR_ERR_INVALID, "NULL argument");
    if (dtype != VR_DTYPE_F32 && dtype != VR_DTYPE_BF16) return fail(VR_ERR_INVALID, "bad dtype %d", dtype);
    VRCHK(set_dev(m->device));
    const std::string name(name_c);
    const int bf = dtype == VR_DTYPE_BF16;
    size_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
    const int E = m->E, I = m->I, V = m->V, QD = m->QD, KVD = m->KVD, QKV = m->QKV;
    auto bad_shape = [&]() { return fail(VR_ERR_INVALID, "unexpected shape for %s", name_c); };
    const bool vis_key = name.rfind("model.visual.", 0) == 0 || name.rfind("visual.", 0) == 0;
    if (vis_key && !m->vis) return VR_OK;                  // no tower attached (vg_vision_create): its tensors are skipped
    Staged st;
    VRCHK(stage(data, numel * (bf ? 2 : 4), on_device, st));
    const void* src = st.dev;
    m->finalized = false;
    VRCHK(end_run(m));
    drop_graph(m);                      // (re)loading a tensor may reallocate it
    if (vis_key) return vision_load_weight(m, name.substr(name.find("visual.") + 7), src, bf, shape, ndim, numel);
    const std::string pre = "model.language_model.";
    if (name == pre + "embed_tokens.weight") {
        if (!shape_is(shape, ndim, {V, E})) return bad_shape();
        VRCHK(m->embed.alloc(numel * 2));
        HIPCHK(launch_pack_weight(src, bf, V, E, E, 0, m->embed.p, E, V, 0, 0, 0));
        HIPCHK(hipDeviceSynchronize());
        m->has_embed = true;
        return VR_OK;
    }
    if (name == pre + "norm.weight") { if (numel != (size_t)E) return bad_shape(); return load_vec(m->final_norm, src, bf, E, E); }
    if (name == "lm_head.weight") { if (!shape_is(shape, ndim, {V, E})) return bad_shape(); return load_linear_part(m->lm_head, V, E, src, bf, V, E, 0, V, 0, 0); }
    if (name.rfind(pre + "layers.", 0) == 0) {
        int n = -1, off = 0;
        if (sscanf(name.c_str(), "model.language_model.layers.%d.%n", &n, &off) < 1) return fail(VR_ERR_INVALID, "bad key %s", name_c);
        if (n < 0 || n >= m->c.num_layers) return fail(VR_ERR_INVALID, "layer index %d out of range", n);
        const std::string sub = name.substr(off);
        GenLayer& l = m->layers[n];
        if (sub == "input_layernorm.weight") { if (numel != (size_t)E) return bad_shape(); return load_vec(l.ln1, src, bf, E, E); }
        if (sub == "post_attention_layernorm.weight") { if (numel != (size_t)E) return bad_shape(); return load_vec(l.ln2, src, bf, E, E); }
        // q | k | v stacked into one [QKV][E] matrix (one GEMM), biases likewise
        static const char* pn[3] = {"self_attn.q_proj.", "self_attn.k_proj.", "self_attn.v_proj."};
        for (int part = 0; part < 3; ++part) {
            const int rows = part == 0 ? QD : KVD, roff = part == 0 ? 0 : (part == 1 ? QD : QD + KVD);
            if (sub == std::string(pn[part]) + "weight") {
                if (!shape_is(shape, ndim, {rows, E})) return bad_shape();
                l.parts_w |= 1 << part;
                return load_linear_part(l.qkv, QKV, E, src, bf, rows, E, 0, rows, 0, roff);
            }
            if (sub == std::string(pn[part]) + "bias") {
                if (numel != (size_t)rows) return bad_shape();
                l.parts_b |= 1 << part;
                return load_bias_part(l.qkv, QKV, src, bf, rows, roff);
            }
        }
        if (sub == "self_attn.o_proj.weight") { if (!shape_is(shape, ndim, {E, QD})) return bad_shape(); return load_linear_part(l.o, E, QD, src, bf, E, QD, 0, E, 0, 0); }
        if (sub == "mlp.gate_proj.weight" || sub == "mlp.up_proj.weight") {
            if (!shape_is(shape, ndim, {I, E})) return bad_shape();
            const int up = sub == "mlp.up_proj.weight";
            l.parts_gu |= 1 << up;
            return load_linear_part(l.gu, 2 * I, E, src, bf, I, E, 0, 16, 32, up * 16);     // 16-row [gate | up] interleave (EPI_SWIGLU)
        }
        if (sub == "mlp.down_proj.weight") { if (!shape_is(shape, ndim, {E, I})) return bad_shape(); return load_linear_part(l.down, E, I, src, bf, E, I, 0, E, 0, 0); }
        return fail(VR_ERR_INVALID, "unknown decoder key %s", name_c);
    }
    return fail(VR_ERR_INVALID, "unknown key %s", name_c);
}

extern "C" int vg_finalize(vg_model_t m) {
    if (!m) return fail(VR_ERR_INVALID, "NULL model");
    if (!m->has_embed || !m->final_norm.ok || !m->lm_head.has_w) return fail(VR_ERR_STATE, "embed_tokens / norm / lm_head missing");
    for (size_t i = 0; i < m->layers.size(); ++i) {
        const GenLayer& l = m->layers[i];
        if (!l.ln1.ok || !l.ln2.ok || l.parts_w != 7 || l.parts_b != 7 || l.parts_gu != 3 || !l.o.has_w || !l.down.has_w)
            return fail(VR_ERR_STATE, "layer %zu is incomplete", i);
    }
    if (m->vis) VRCHK(vision_check_complete(m));
    VRCHK(set_dev(m->device));
    VRCHK(persist_setup(m));
    m->finalized = true;
    return VR_OK;
}

// one decoder layer over T rows that already sit (normalised, bf16) in w_xn; cache rows [len, len + T)
static int gen_layer(vg_model_s* m, int l, int T, bool decode, const float* next_norm, hipStream_t s) {
    const vg_config_t& c = m->c;
    GenLayer& L = m->layers[l];
    const int E = m->E, QD = m->QD, QKV = m->QKV, Ip = pad128(m->I);
    float* h = m->w_h.as<float>();
    float* part = m->w_part.as<float>();
    int* cu = m->w_cu.as<int>();
    // ---- q | k | v
    if (decode) {
        GemmArgs a = gen_gemm_args(m->w_xn.p, E, L.qkv, T, part, QKV);
        a.bias = nullptr;
        a.ksplit = choose_ksplit(QKV, E, VR_KS_QKV);
        a.split_stride = (size_t)QKV * T;
        HIPCHK(launch_gemm_skinny(a, s));
        GenState* st = m->w_state.as<GenState>();          // position and cache row of the step: on the device
        HIPCHK(launch_mrope_cache(nullptr, part, a.ksplit, (size_t)QKV * T, L.qkv.b.as<float>(), QKV, T, m->H, m->KV, st->pos,
                                  1, c.mrope_section[0], c.mrope_section[1], m->inv_freq.as<float>(), m->w_q.p, QD,
                                  gen_kc(m, l, m->cur), gen_vc(m, l, m->cur), m->KVD, 0, nullptr, s, &st->len));
    } else {
        GemmArgs a = gen_gemm_args(m->w_xn.p, E, L.qkv, T, m->w_qkv.p, QKV);
        HIPCHK(launch_gemm(a, EPI_BF16, GEMM_VARIANT_AUTO, s));
        HIPCHK(launch_mrope_cache(m->w_qkv.p, nullptr, 0, 0, nullptr, QKV, T, m->H, m->KV, m->w_pos.as<int>(), m->Tcap,
                                  c.mrope_section[0], c.mrope_section[1], m->inv_freq.as<float>(), m->w_q.p, QD, gen_kc(m, l, m->cur),
                                  gen_vc(m, l, m->cur), m->KVD, m->len, nullptr, s));
    }
    // ---- grouped-query attention over the cache.  Prefill: causal within the prompt.  Decode: the new row sees the
    //      whole cache; one query row x 28 heads would be 28 workgroups, so the cache is cut into KV ranges
    //      (decode_begin_kernel, GenState::cu_kv) that run as independent "sequences" sharing the query row (q_shared) —
    //      always GEN_ATT_SPLITS of them on the grid, the ones past the end empty — and a small kernel merges the
    //      GenState::splits real ones by their log-sum-exps.  Nothing here depends on a host-side length.
    {
        AttnArgs a{};
        a.q = m->w_q.p; a.ldq = QD; a.k = gen_kc(m, l, m->cur); a.ldk = m->KVD; a.v = gen_vc(m, l, m->cur); a.ldv = m->KVD;
        a.heads = m->H; a.head_dim = 128; a.scale = 1.0f / sqrtf(128.0f); a.kv_group = m->H / m->KV;
        if (decode) {
            // the G = H / KV query heads that share a KV head are the ROWS of one tile (q_head_stride): a KV range is read
            // once per group instead of once per query head, on KV x ranges workgroups
            const int G = m->H / m->KV;
            GenState* st = m->w_state.as<GenState>();
            a.cu_q = st->cu_q; a.cu_kv = st->cu_kv;
            a.heads = m->KV; a.kv_group = 1; a.ldq = 128; a.q_head_stride = G * 128;
            a.out = m->w_attp.p; a.ldo = m->KVD; a.B = GEN_ATT_SPLITS; a.max_q = G; a.causal = 0; a.q_shared = 1;
            a.lse = m->w_lse.as<float>();
            HIPCHK(launch_attention(a, s));
            // (the ranges are merged by the o projection below as it builds its A row: no launch of its own)
        } else {
            a.cu_q = cu; a.cu_kv = cu + 2;
            a.out = m->w_att.p; a.ldo = QD; a.B = 1; a.max_q = T; a.causal = 1; a.q_shared = 0;
            HIPCHK(launch_attention(a, s));
        }
    }
    // ---- o projection + residual, post-attention norm
    if (decode) {
        GemmArgs a = gen_gemm_args(m->w_att.p, QD, L.o, T, part, E);
        a.ksplit = choose_ksplit(E, QD, VR_KS_O);
        a.split_stride = (size_t)E * T;
        const int per = (QD / 64 + a.ksplit - 1) / a.ksplit;
        if (T == 1 && per <= 4) {
            GenState* st = m->w_state.as<GenState>();
            SkinnyCombine cb{m->w_attp.p, m->w_lse.as<float>(), 0, m->H, m->H / m->KV, &st->splits};
            HIPCHK(launch_gemm_skinny(a, s, false, &cb));
        } else {                                    // (a K range longer than the stage ring: merge with its own launch)
            GenState* st = m->w_state.as<GenState>();
            HIPCHK(launch_attn_combine(m->w_attp.p, m->w_lse.as<float>(), 0, m->H, m->H / m->KV, m->w_att.p, s, &st->splits));
            HIPCHK(launch_gemm_skinny(a, s));
        }
        HIPCHK(launch_rmsnorm_accum(h, T, E, E, part, a.ksplit, (size_t)E * T, E, 1.0f, L.ln2.v.as<float>(), c.rms_norm_eps, m->w_xn.p, E, s));
    } else {
        GemmArgs a = gen_gemm_args(m->w_att.p, QD, L.o, T, h, E);
        a.resid = h;
        HIPCHK(launch_gemm(a, EPI_RESID, GEMM_VARIANT_AUTO, s));
        HIPCHK(launch_rmsnorm(h, T, E, E, L.ln2.v.as<float>(), c.rms_norm_eps, m->w_xn.p, E, s));
    }
    // ---- SwiGLU MLP + residual; the next layer's (or the final) norm closes the layer
    if (decode) {
        const int N2 = L.gu.n_pad;
        GemmArgs g = gen_gemm_args(m->w_xn.p, E, L.gu, T, part, N2);
        g.ksplit = choose_ksplit(N2, E, VR_KS_GU);
        if (g.ksplit == 1) {                        // the usual case (more tiles than a split would help): SwiGLU in the tile's epilogue
            g.out = m->w_act.p; g.ldo = Ip;
            HIPCHK(launch_gemm_skinny(g, s, true));
        } else {
            g.split_stride = (size_t)N2 * T;
            HIPCHK(launch_gemm_skinny(g, s));
            HIPCHK(launch_swiglu_sum(part, g.ksplit, (size_t)N2 * T, N2, T, m->I, m->w_act.p, Ip, s));
        }
        GemmArgs a = gen_gemm_args(m->w_act.p, Ip, L.down, T, part, E);
        a.ksplit = choose_ksplit(E, L.down.k_pad, VR_KS_DOWN);
        a.split_stride = (size_t)E * T;
        HIPCHK(launch_gemm_skinny(a, s));
        HIPCHK(launch_rmsnorm_accum(h, T, E, E, part, a.ksplit, (size_t)E * T, E, 1.0f, next_norm, c.rms_norm_eps, m->w_xn.p, E, s));
    } else {
        { GemmArgs a = gen_gemm_args(m->w_xn.p, E, L.gu, T, m->w_act.p, Ip); HIPCHK(launch_gemm(a, EPI_SWIGLU, GEMM_VARIANT_AUTO, s)); }
        GemmArgs a = gen_gemm_args(m->w_act.p, Ip, L.down, T, h, E);
        a.resid = h;
        HIPCHK(launch_gemm(a, EPI_RESID, GEMM_VARIANT_AUTO, s));
        if (l + 1 < (int)m->layers.size()) HIPCHK(launch_rmsnorm(h, T, E, E, next_norm, c.rms_norm_eps, m->w_xn.p, E, s));
    }
    return VR_OK;
}

// final norm of one row (already in w_xn row 0 when `normed`) + lm_head -> w_logits
static int gen_head(vg_model_s* m, const float* h_row, bool normed, hipStream_t s) {
    const int E = m->E;
    const void* A = m->w_xn.p;
    if (!normed) {
        HIPCHK(launch_rmsnorm(h_row, 1, E, E, m->final_norm.v.as<float>(), m->c.rms_norm_eps, m->w_last.p, E, s));
        A = m->w_last.p;
    }
    GemmArgs a = gen_gemm_args(A, E, m->lm_head, 1, gen_logits(m, m->cur), m->lm_head.n_pad);
    HIPCHK(launch_gemm_skinny(a, s));
    m->have_logits = true;
    return VR_OK;
}

extern "C" int vg_prefill(vg_model_t m, const int32_t* ids, int32_t T, const int32_t* embed_rows, const float* embeds,
                          int32_t n_embed, const int32_t* pos3, void* stream) {
    if (!m || !ids || !pos3) return fail(VR_ERR_INVALID, "NULL argument");
    if (!m->finalized) return fail(VR_ERR_STATE, "vg_finalize has not succeeded");
    if (T <= 0 || T > m->c.max_prefill) return fail(VR_ERR_CAPACITY, "%d prompt tokens (max_prefill %d)", T, m->c.max_prefill);
    if (n_embed < 0 || n_embed > T || (n_embed > 0 && !embed_rows)) return fail(VR_ERR_INVALID, "bad embedding overrides");
    if (n_embed > 0 && !embeds && n_embed != m->vis_tokens)
        return fail(VR_ERR_STATE, "embeds is NULL: %d rows asked for, the last vg_vision_encode left %d on the device", n_embed, m->vis_tokens);
    for (int i = 0; i < T; ++i)
        if (ids[i] < 0 || ids[i] >= m->V) return fail(VR_ERR_INVALID, "token id %d out of range", ids[i]);
    for (int i = 0; i < n_embed; ++i)
        if (embed_rows[i] < 0 || embed_rows[i] >= T) return fail(VR_ERR_INVALID, "embedding row %d out of range", embed_rows[i]);
    VRCHK(set_dev(m->device));
    VRCHK(end_run(m));
    hipStream_t s = (hipStream_t)stream;
    const int E = m->E;
    m->len = 0;
    m->have_logits = false;
    m->tok_on_device = false;
    HIPCHK(hipMemsetAsync(gen_seen(m, m->cur), 0, (size_t)((m->V + 31) / 32) * 4, s));
    HIPCHK(hipMemcpyAsync(m->w_ids.p, ids, (size_t)T * 4, hipMemcpyHostToDevice, s));
    for (int c = 0; c < 3; ++c)
        HIPCHK(hipMemcpyAsync(m->w_pos.as<int>() + (size_t)c * m->Tcap, pos3 + (size_t)c * T, (size_t)T * 4, hipMemcpyHostToDevice, s));
    const int cu_host[4] = {0, T, 0, T};              // cu_q, cu_kv
    HIPCHK(hipMemcpyAsync(m->w_cu.p, cu_host, sizeof(cu_host), hipMemcpyHostToDevice, s));
    HIPCHK(launch_mark_seen(m->w_ids.as<int>(), T, gen_seen(m, m->cur), m->V, s));
    HIPCHK(launch_embed_gather(m->w_ids.as<int>(), T, m->embed.p, E, 1.0f, m->w_h.as<float>(), s));
    if (n_embed > 0) {
        HIPCHK(hipMemcpyAsync(m->w_erows.p, embed_rows, (size_t)n_embed * 4, hipMemcpyHostToDevice, s));
        if (embeds) {
            HIPCHK(hipMemcpyAsync(m->w_emb.p, embeds, (size_t)n_embed * E * 4, hipMemcpyHostToDevice, s));
            m->vis_tokens = 0;                                // w_emb no longer holds a tower result
        }
        HIPCHK(launch_scatter_rows(m->w_emb.as<float>(), m->w_erows.as<int>(), n_embed, E, m->w_h.as<float>(), E, s));
    }
    HIPCHK(hipStreamSynchronize(s));            // the host buffers above may go away after the call
    HIPCHK(launch_rmsnorm(m->w_h.as<float>(), T, E, E, m->layers[0].ln1.v.as<float>(), m->c.rms_norm_eps, m->w_xn.p, E, s));
    const int nl = (int)m->layers.size();
    for (int l = 0; l < nl; ++l)
        VRCHK(gen_layer(m, l, T, false, l + 1 < nl ? m->layers[l + 1].ln1.v.as<float>() : nullptr, s));
    m->len = T;
    return gen_head(m, m->w_h.as<float>() + (size_t)(T - 1) * E, false, s);
}

// One decode step on stream s: everything it needs beyond the weights and caches is in GenState on the device, so the
// same sequence of launches serves a host-driven step (vg_decode uploads token / positions / length first) and the
// captured graph of a free-running step (sampled: + the sampling kernels, which also advance the state).
static int enqueue_decode(vg_model_s* m, hipStream_t s, bool sampled, float temperature, float penalty, unsigned long long seed) {
    const int E = m->E;
    GenState* st = m->w_state.as<GenState>();
    HIPCHK(launch_decode_begin(st, m->H / m->KV, s));
    HIPCHK(launch_embed_gather(&st->token, 1, m->embed.p, E, 1.0f, m->w_h.as<float>(), s));
    HIPCHK(launch_rmsnorm(m->w_h.as<float>(), 1, E, E, m->layers[0].ln1.v.as<float>(), m->c.rms_norm_eps, m->w_xn.p, E, s));
    const int nl = (int)m->layers.size();
    if (m->p_grid > 0) {
        VRCHK(persist_layers(m, s));               // all layers in one launch; leaves the lm_head's A row in w_xn like the loop below
    } else {
        for (int l = 0; l < nl; ++l)
            VRCHK(gen_layer(m, l, 1, true, l + 1 < nl ? m->layers[l + 1].ln1.v.as<float>() : m->final_norm.v.as<float>(), s));
    }
    VRCHK(gen_head(m, nullptr, true, s));
    if (sampled)
        HIPCHK(launch_sample(gen_logits(m, m->cur), m->V, gen_seen(m, m->cur), penalty, temperature, seed, 0, m->w_tok.as<int>(),
                             m->w_tok.as<unsigned long long>() + 2, s, st, 1, 1));
    return VR_OK;
}

// a free run still has steps in flight: finish them before anything else touches the model
static int end_run(vg_model_s* m) {
    if (!m->running) return VR_OK;
    m->running = false;
    HIPCHK(hipStreamSynchronize(m->run_on));
    return VR_OK;
}

extern "C" int vg_decode(vg_model_t m, int32_t token, const int32_t pos[3], void* stream) {
    if (!m || !pos) return fail(VR_ERR_INVALID, "NULL argument");
    if (!m->finalized || m->len <= 0) return fail(VR_ERR_STATE, "no sequence in progress (vg_prefill first)");
    if (m->len >= m->c.max_len) return fail(VR_ERR_CAPACITY, "KV cache is full (%d rows)", m->c.max_len);
    if (token < 0 || token >= m->V) return fail(VR_ERR_INVALID, "token id %d out of range", token);
    VRCHK(set_dev(m->device));
    VRCHK(end_run(m));
    hipStream_t s = (hipStream_t)stream;
    const int head[5] = {token, pos[0], pos[1], pos[2], m->len};         // GenState::token, pos[3], len
    HIPCHK(hipMemcpyAsync(m->w_state.p, head, sizeof(head), hipMemcpyHostToDevice, s));
    m->tok_on_device = false;
    VRCHK(enqueue_decode(m, s, false, 0.f, 1.f, 0));
    m->len += 1;
    return VR_OK;
}

extern "C" int vg_sample(vg_model_t m, float temperature, float repetition_penalty, uint64_t seed, int32_t step,
                         int32_t* token_out, void* stream) {
    if (!m || !token_out) return fail(VR_ERR_INVALID, "NULL argument");
    if (!m->have_logits) return fail(VR_ERR_STATE, "no logits yet (vg_prefill / vg_decode first)");
    if (!(repetition_penalty > 0.f) || temperature < 0.f) return fail(VR_ERR_INVALID, "bad sampling parameters");
    VRCHK(set_dev(m->device));
    VRCHK(end_run(m));
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(launch_sample(gen_logits(m, m->cur), m->V, gen_seen(m, m->cur), repetition_penalty, temperature, seed,
                         (unsigned)step, m->w_tok.as<int>(), m->w_tok.as<unsigned long long>() + 2, s, m->w_state.as<GenState>(), 0, 0));
    HIPCHK(hipMemcpyAsync(token_out, m->w_tok.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    m->tok_on_device = true;
    return persist_check(m);
}

// ---- free-running generation ------------------------------------------------------------------------------------
static int capture_step(vg_model_s* m, float temperature, float penalty, unsigned long long seed) {
    // (vg_select drops the graph when the slot changes: it bakes the slot's cache / logits / seen pointers in)
    if (m->graph_exec && m->g_temp == temperature && m->g_pen == penalty && m->g_seed == seed) return VR_OK;
    drop_graph(m);
    HIPCHK(hipStreamBeginCapture(m->run_stream, hipStreamCaptureModeRelaxed));
    const int rc = enqueue_decode(m, m->run_stream, true, temperature, penalty, seed);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(m->run_stream, &g);       // (always: the stream must leave capture mode)
    if (rc != VR_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess || !g) return fail(VR_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    m->graph = g;
    HIPCHK(hipGraphInstantiate(&m->graph_exec, m->graph, nullptr, nullptr, 0));
    m->g_temp = temperature; m->g_pen = penalty; m->g_seed = seed;
    return VR_OK;
}

extern "C" int vg_run_begin(vg_model_t m, int32_t position, float temperature, float repetition_penalty, uint64_t seed,
                            int32_t first_step, void* stream) {
    if (!m) return fail(VR_ERR_INVALID, "NULL model");
    if (!m->finalized || m->len <= 0) return fail(VR_ERR_STATE, "no sequence in progress (vg_prefill first)");
    if (!m->tok_on_device) return fail(VR_ERR_STATE, "no sampled token on the device (vg_sample first)");
    if (!(repetition_penalty > 0.f) || temperature < 0.f) return fail(VR_ERR_INVALID, "bad sampling parameters");
    VRCHK(set_dev(m->device));
    VRCHK(end_run(m));
    if (!m->run_stream) {
        HIPCHK(hipStreamCreate(&m->run_stream));        // blocking: ordered against the default stream like any other
        HIPCHK(hipHostMalloc((void**)&m->h_tokens, GEN_RUN_RING * sizeof(int), hipHostMallocDefault));
        for (auto& e : m->run_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    hipStream_t s = stream ? (hipStream_t)stream : m->run_stream;
    HIPCHK(hipStreamSynchronize(s));
    VRCHK(capture_step(m, temperature, repetition_penalty, seed));
    const int body[5] = {position, position, position, m->len, first_step};          // GenState::pos[3], len, step
    HIPCHK(hipMemcpyAsync((char*)m->w_state.p + offsetof(GenState, pos), body, sizeof(body), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    m->running = true; m->run_on = s; m->run_steps = 0;
    return VR_OK;
}

extern "C" int vg_run_step(vg_model_t m) {
    if (!m) return fail(VR_ERR_INVALID, "NULL model");
    if (!m->running) return fail(VR_ERR_STATE, "no free run in progress (vg_run_begin)");
    if (m->len >= m->c.max_len) return fail(VR_ERR_CAPACITY, "KV cache is full (%d rows)", m->c.max_len);
    VRCHK(set_dev(m->device));
    const int slot = m->run_steps % GEN_RUN_RING;
    if (m->run_steps >= GEN_RUN_RING) HIPCHK(hipEventSynchronize(m->run_ev[slot]));     // the slot's previous token has landed
    HIPCHK(hipGraphLaunch(m->graph_exec, m->run_on));
    HIPCHK(hipMemcpyAsync(m->h_tokens + slot, m->w_tok.p, 4, hipMemcpyDeviceToHost, m->run_on));
    HIPCHK(hipEventRecord(m->run_ev[slot], m->run_on));
    m->len += 1;
    m->run_steps += 1;
    m->have_logits = true;
    return VR_OK;
}

extern "C" int vg_run_token(vg_model_t m, int32_t index, int32_t* token) {
    if (!m || !token) return fail(VR_ERR_INVALID, "NULL argument");
    if (index < 0 || index >= m->run_steps || index < m->run_steps - GEN_RUN_RING)
        return fail(VR_ERR_INVALID, "step %d is not among the last %d of %d enqueued", index, GEN_RUN_RING, m->run_steps);
    VRCHK(set_dev(m->device));
    HIPCHK(hipEventSynchronize(m->run_ev[index % GEN_RUN_RING]));
    *token = m->h_tokens[index % GEN_RUN_RING];
    return persist_check(m);
}

extern "C" int vg_run_end(vg_model_t m) {
    if (!m) return fail(VR_ERR_INVALID, "NULL model");
    VRCHK(set_dev(m->device));
    return end_run(m);
}

// ---- several sequences ---------------------------------------------------------------------------------------------
extern "C" int vg_select(vg_model_t m, int32_t slot) {
    if (!m) return fail(VR_ERR_INVALID, "NULL model");
    if (slot < 0 || slot >= m->n_slots) return fail(VR_ERR_INVALID, "slot %d: the model holds %d sequence(s) (vg_config_t::max_seqs)", slot, m->n_slots);
    VRCHK(set_dev(m->device));
    VRCHK(end_run(m));
    if (slot == m->cur) return VR_OK;
    m->slot_len[m->cur] = m->len; m->slot_logits[m->cur] = m->have_logits;
    m->cur = slot;
    m->len = m->slot_len[slot]; m->have_logits = m->slot_logits[slot] != 0;
    m->tok_on_device = false;
    drop_graph(m);
    return persist_setup(m);                       // the layer table points at the slot's caches
}

// one decoder layer of a batched decode step: n rows, one per sequence, each with its own cache rows
static int gen_layer_batch(vg_model_s* m, int l, int n, const float* next_norm, hipStream_t s) {
    const vg_config_t& c = m->c;
    GenLayer& L = m->layers[l];
    const int E = m->E, QD = m->QD, QKV = m->QKV, Ip = pad128(m->I), G = m->H / m->KV;
    float* h = m->w_h.as<float>();
    float* part = m->w_part.as<float>();
    GenBatch* bt = m->w_batch.as<GenBatch>();
    {   // q | k | v: one pass over the weights for the n rows, then rotate + append to each sequence's cache
        GemmArgs a = gen_gemm_args(m->w_xn.p, E, L.qkv, n, part, QKV);
        a.bias = nullptr;
        a.ksplit = choose_ksplit(QKV, E, VR_KS_QKV);
        a.split_stride = (size_t)QKV * n;
        HIPCHK(launch_gemm_skinny(a, s));
        HIPCHK(launch_mrope_cache(nullptr, part, a.ksplit, (size_t)QKV * n, L.qkv.b.as<float>(), QKV, n, m->H, m->KV, &bt->pos[0][0], 16,
                                  c.mrope_section[0], c.mrope_section[1], m->inv_freq.as<float>(), m->w_q.p, QD, m->kc[l].p, m->vc[l].p,
                                  m->KVD, 0, nullptr, s, nullptr, bt->cache_row));
    }
    {   // attention: item (r, range) = the group's query heads of row r against one KV range of ITS cache
        AttnArgs a{};
        a.q = m->w_q.p; a.ldq = 128; a.q_head_stride = G * 128; a.q_in_rows = bt->q_in;
        a.k = m->kc[l].p; a.ldk = m->KVD; a.v = m->vc[l].p; a.ldv = m->KVD;
        a.cu_q = bt->cu_q; a.cu_kv = bt->kv_lo; a.kv_end = bt->kv_hi;
        a.heads = m->KV; a.kv_group = 1; a.head_dim = 128; a.scale = 1.0f / sqrtf(128.0f);
        a.out = m->w_attp.p; a.ldo = m->KVD; a.B = n * GEN_ATT_SPLITS; a.max_q = G; a.causal = 0; a.q_shared = 0;
        a.lse = m->w_lse.as<float>();
        HIPCHK(launch_attention(a, s));
        HIPCHK(launch_attn_combine(m->w_attp.p, m->w_lse.as<float>(), 0, m->H, G, m->w_att.p, s, bt->splits, n, QD));
    }
    {
        GemmArgs a = gen_gemm_args(m->w_att.p, QD, L.o, n, part, E);
        a.ksplit = choose_ksplit(E, QD, VR_KS_O);
        a.split_stride = (size_t)E * n;
        HIPCHK(launch_gemm_skinny(a, s));
        HIPCHK(launch_rmsnorm_accum(h, n, E, E, part, a.ksplit, (size_t)E * n, E, 1.0f, L.ln2.v.as<float>(), c.rms_norm_eps, m->w_xn.p, E, s));
    }
    {
        const int N2 = L.gu.n_pad;
        GemmArgs g = gen_gemm_args(m->w_xn.p, E, L.gu, n, part, N2);
        g.ksplit = choose_ksplit(N2, E, VR_KS_GU);
        if (g.ksplit == 1) {
            g.out = m->w_act.p; g.ldo = Ip;
            HIPCHK(launch_gemm_skinny(g, s, true));
        } else {
            g.split_stride = (size_t)N2 * n;
            HIPCHK(launch_gemm_skinny(g, s));
            HIPCHK(launch_swiglu_sum(part, g.ksplit, (size_t)N2 * n, N2, n, m->I, m->w_act.p, Ip, s));
        }
        GemmArgs a = gen_gemm_args(m->w_act.p, Ip, L.down, n, part, E);
        a.ksplit = choose_ksplit(E, L.down.k_pad, VR_KS_DOWN);
        a.split_stride = (size_t)E * n;
        HIPCHK(launch_gemm_skinny(a, s));
        HIPCHK(launch_rmsnorm_accum(h, n, E, E, part, a.ksplit, (size_t)E * n, E, 1.0f, next_norm, c.rms_norm_eps, m->w_xn.p, E, s));
    }
    return VR_OK;
}

extern "C" int vg_decode_batch(vg_model_t m, int32_t n, const int32_t* slots, const int32_t* tokens, const int32_t* pos, void* stream) {
    if (!m || !slots || !tokens || !pos) return fail(VR_ERR_INVALID, "NULL argument");
    if (!m->finalized) return fail(VR_ERR_STATE, "vg_finalize has not succeeded");
    if (n <= 0 || n > m->n_slots || n > 16) return fail(VR_ERR_INVALID, "%d rows: the model holds %d sequence(s), a step takes at most 16", n, m->n_slots);
    VRCHK(set_dev(m->device));
    VRCHK(end_run(m));
    m->slot_len[m->cur] = m->len; m->slot_logits[m->cur] = m->have_logits;
    const int G = m->H / m->KV;
    {   // the widest split-K plane set of the step must fit the partial buffer
        const size_t cap = m->w_part.bytes / 4;
        const size_t need = (size_t)n * std::max({(size_t)choose_ksplit(m->QKV, m->E, VR_KS_QKV) * m->QKV, (size_t)choose_ksplit(m->E, m->QD, VR_KS_O) * m->E,
                                                  (size_t)choose_ksplit(m->E, pad128(m->I), VR_KS_DOWN) * m->E});
        if (need > cap) return fail(VR_ERR_CAPACITY, "%d rows need %zu partial-sum floats (%zu available)", n, need, cap);
    }
    GenBatch b{};
    unsigned used = 0;
    for (int r = 0; r < n; ++r) {
        const int sl = slots[r];
        if (sl < 0 || sl >= m->n_slots || (used >> sl) & 1u) return fail(VR_ERR_INVALID, "row %d: slot %d is out of range or named twice", r, sl);
        used |= 1u << sl;
        const int len = m->slot_len[sl];
        if (len <= 0) return fail(VR_ERR_STATE, "slot %d has no sequence in progress (vg_select + vg_prefill first)", sl);
        if (len >= m->c.max_len) return fail(VR_ERR_CAPACITY, "slot %d: KV cache is full (%d rows)", sl, m->c.max_len);
        if (tokens[r] < 0 || tokens[r] >= m->V) return fail(VR_ERR_INVALID, "token id %d out of range", tokens[r]);
        b.token[r] = tokens[r];
        for (int c = 0; c < 3; ++c) b.pos[c][r] = pos[3 * r + c];
        const int base = sl * m->c.max_len;
        b.cache_row[r] = base + len;
        // KV ranges of the row's attention: decode_begin_kernel's split of its L = len + 1 cache rows
        const int L = len + 1;
        int splits = std::min(GEN_ATT_SPLITS, std::max(1, (L + 127) / 128));
        const int chunk = ((L + splits - 1) / splits + 63) / 64 * 64;
        splits = (L + chunk - 1) / chunk;
        b.splits[r] = splits;
        for (int t = 0; t < GEN_ATT_SPLITS; ++t) {
            const int it = r * GEN_ATT_SPLITS + t;
            b.kv_lo[it] = base + std::min(L, t * chunk);
            b.kv_hi[it] = base + std::min(L, (t + 1) * chunk);
            b.q_in[it] = r * m->H;                       // q rows are 128 wide: row r's heads start at r * H
        }
    }
    for (int it = 0; it <= n * GEN_ATT_SPLITS; ++it) b.cu_q[it] = it * G;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(m->w_batch.p, &b, sizeof(b), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));                     // `b` lives on this frame
    GenBatch* bt = m->w_batch.as<GenBatch>();
    const int E = m->E;
    HIPCHK(launch_embed_gather(bt->token, n, m->embed.p, E, 1.0f, m->w_h.as<float>(), s));
    HIPCHK(launch_rmsnorm(m->w_h.as<float>(), n, E, E, m->layers[0].ln1.v.as<float>(), m->c.rms_norm_eps, m->w_xn.p, E, s));
    const int nl = (int)m->layers.size();
    for (int l = 0; l < nl; ++l)
        VRCHK(gen_layer_batch(m, l, n, l + 1 < nl ? m->layers[l + 1].ln1.v.as<float>() : m->final_norm.v.as<float>(), s));
    {   // lm_head over the n normed rows, then every row to its slot's logits
        GemmArgs a = gen_gemm_args(m->w_xn.p, E, m->lm_head, n, m->w_logits_b.p, m->lm_head.n_pad);
        HIPCHK(launch_gemm_skinny(a, s));
        for (int r = 0; r < n; ++r)
            HIPCHK(hipMemcpyAsync(gen_logits(m, slots[r]), m->w_logits_b.as<float>() + (size_t)r * m->lm_head.n_pad, (size_t)m->V * 4,
                                  hipMemcpyDeviceToDevice, s));
    }
    for (int r = 0; r < n; ++r) { m->slot_len[slots[r]] += 1; m->slot_logits[slots[r]] = 1; }
    m->len = m->slot_len[m->cur]; m->have_logits = m->slot_logits[m->cur] != 0;
    m->tok_on_device = false;
    return VR_OK;
}

extern "C" int vg_sample_batch(vg_model_t m, int32_t n, const int32_t* slots, float temperature, float repetition_penalty, uint64_t seed,
                               int32_t step, int32_t* tokens_out, void* stream) {
    if (!m || !slots || !tokens_out) return fail(VR_ERR_INVALID, "NULL argument");
    if (n <= 0 || n > m->n_slots || n > 16) return fail(VR_ERR_INVALID, "%d rows: the model holds %d sequence(s)", n, m->n_slots);
    if (!(repetition_penalty > 0.f) || temperature < 0.f) return fail(VR_ERR_INVALID, "bad sampling parameters");
    VRCHK(set_dev(m->device));
    VRCHK(end_run(m));
    m->slot_logits[m->cur] = m->have_logits;
    for (int r = 0; r < n; ++r)
        if (slots[r] < 0 || slots[r] >= m->n_slots || !m->slot_logits[slots[r]]) return fail(VR_ERR_STATE, "slot %d has no logits", slots[r]);
    hipStream_t s = (hipStream_t)stream;
    int* toks = m->w_batch.as<GenBatch>()->sampled;
    for (int r = 0; r < n; ++r)
        HIPCHK(launch_sample(gen_logits(m, slots[r]), m->V, gen_seen(m, slots[r]), repetition_penalty, temperature, seed, (unsigned)step,
                             toks + r, m->w_tok.as<unsigned long long>() + 2, s, nullptr, 0, 0));
    HIPCHK(hipMemcpyAsync(tokens_out, toks, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    m->tok_on_device = false;
    return VR_OK;
}

extern "C" int vg_logits(vg_model_t m, float* out, void* stream) {
    if (!m || !out) return fail(VR_ERR_INVALID, "NULL argument");
    if (!m->have_logits) return fail(VR_ERR_STATE, "no logits yet");
    VRCHK(set_dev(m->device));
    VRCHK(end_run(m));
    HIPCHK(hipMemcpyAsync(out, gen_logits(m, m->cur), (size_t)m->V * 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return persist_check(m);
}

extern "C" int vg_cache_len(vg_model_t m, int32_t* len) {
    if (!m || !len) return fail(VR_ERR_INVALID, "NULL argument");
    *len = m->len;
    return VR_OK;
}
