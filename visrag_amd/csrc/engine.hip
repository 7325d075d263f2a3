// Host side of libvisrag_hip.so: the C ABI of include/visrag_hip.h — device weight store,
// workspace, per-grid tables, and the launch sequence of one VisRAG-Ret encode pass.
// Reference call stack this replaces: SURVEY.md section 3.1 (modeling_visrag_ret.py:86-126 ->
// modeling_minicpmv.py:95-171 -> vision_transformer.py:682-692 / resampler.py:146-168 ->
// modeling_minicpm.py:1147-1304 -> dense_retrieval_model.py:180-184,222-223).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <string>
#include <vector>

#include "engine_common.h"

extern "C" const char* vr_version(void) { return "visrag_hip 0.1.0 (gfx950)"; }
extern "C" const char* vr_last_error(void) { return g_err.c_str(); }
extern "C" int vr_device_count(int* count) {
    if (!count) return fail(VR_ERR_INVALID, "count is NULL");
    HIPCHK(hipGetDeviceCount(count));
    return VR_OK;
}


struct VitBlock {
    Vec n1w, n1b, n2w, n2b; Linear qkv, proj, fc1, fc2;
};
struct DecLayer {
    Vec ln1, ln2; Linear qkv, o, gu, down; int parts_qkv = 0, parts_gu = 0;
    Linear qkv_lo, o_lo, gu_lo, down_lo;     // w - bf16(w) of fp32 source weights: the split-precision text path (hp_text.hip)
};

struct GridTables {
    int gh = 0, gw = 0;
    DevBuf vit_pos;        // f32 [N][Dp]  : bicubic-antialias resample of vpm.pos_embed
    DevBuf pos_k;          // f32 [N][E]   : sincos2d(N,E) @ Wk^T (the k-side position term)
};

struct Tap { std::vector<float> data; int64_t rows = 0, cols = 0; };

struct vr_model_s {
    int device = 0;
    vr_config_t c{};
    bool finalized = false, taps_on = false;
    int pool_mode = 0;                        // VR_POOL_*
    bool borrowed = false;                    // vr_model_clone: weights belong to another handle
    // dims
    int D = 0, Dp = 0, F = 0, Fp = 0, E = 0, I = 0, Ip = 0, Kpe = 0, Kpe_p = 0, Q = 0;
    // weights
    Linear patch;
    std::vector<float> pos_embed_host;       // [G*G][D]
    std::vector<VitBlock> blocks;
    Vec vit_nw, vit_nb;
    Linear r_kvproj, r_kv, r_out, r_proj;    // r_kv = in_proj rows [E,3E) (k|v)
    Vec r_lnq_w, r_lnq_b, r_lnkv_w, r_lnkv_b, r_lnpost_w, r_lnpost_b;
    std::vector<float> r_query_host, r_wq_host, r_bq_host;   // for the one-time q projection
    bool has_query = false, has_inproj = false, has_inproj_b = false, has_pos = false;
    DevBuf r_q;                               // bf16 [64][E] projected queries
    DevBuf embed;  bool has_embed = false;    // bf16 [V][E]
    DevBuf embed_lo; bool has_embed_lo = false;   // its low half (fp32 source, split-precision text path)
    std::vector<DecLayer> layers;
    Vec final_norm;
    DevBuf rope;                              // f32 [max_pos][64]
    int rope_len = 0;
    std::map<std::pair<int, int>, GridTables> grids;
    // workspace
    int64_t Mcap = 0, Tcap = 0, Rcap = 0;     // padded rows: patches, tokens, resampler rows
    DevBuf w_hvit, w_xn, w_qkv, w_att, w_mlp, w_kv32, w_xkv, w_KV, w_ratt, w_rout, w_rln;
    DevBuf w_h, w_dxn, w_dqkv, w_datt, w_dact, w_part;   // w_part: split-K partial products [3][T][E] f32
    DevBuf w_cu, w_ids, w_seq, w_pos, w_rowmap, w_imgptr, w_pix, w_out;
    DevBuf w_hp_hi, w_hp_planes, w_hp_qkv, w_hp_att, w_hp_gu, w_seqof;   // split-precision text path (hp_text.hip)
    DevBuf w_hp_part;                                                 // its split-K planes for short batches (grown on demand)
    DevBuf w_hidden;                                                  // vr_encode_hidden: packed post-norm rows when the resampler's scratch is too small (grown on demand)
    std::map<std::string, Tap> taps;
    // HIP-event profiling of kernel classes (bench.py roofline): pairs recorded on the launch
    // stream, elapsed times summed lazily in vr_model_get_profile
    bool prof_on = false;
    int prof_level = 0;                       // 1: the seven phase classes; 2: the decoder's sub-phases instead (events between its kernels)
    struct ProfClass { std::vector<hipEvent_t> ev; size_t used = 0; double ms = 0, flops = 0; int64_t launches = 0; };
    ProfClass prof[VR_PROF_CLASSES];
    // pinned host arena for the small per-call arrays (ids, offsets, row maps, image pointers):
    // async H2D copies read it after vr_encode returned, `arena_ev` marks when they have run.
    char* arena = nullptr; size_t arena_cap = 0, arena_used = 0;
    hipEvent_t arena_ev = nullptr; bool arena_pending = false, arena_open = false;
};

static int arena_begin(vr_model_s* m, size_t need) {
    if (m->arena_pending) { HIPCHK(hipEventSynchronize(m->arena_ev)); m->arena_pending = false; }
    if (!m->arena_ev) HIPCHK(hipEventCreateWithFlags(&m->arena_ev, hipEventDisableTiming));
    if (m->arena_cap < need) {
        if (m->arena) (void)hipHostFree(m->arena);
        m->arena = nullptr; m->arena_cap = 0;
        const size_t cap = std::max(need, (size_t)4 << 20);
        HIPCHK(hipHostMalloc((void**)&m->arena, cap, hipHostMallocDefault));
        m->arena_cap = cap;
    }
    m->arena_used = 0;
    m->arena_open = true;        // async copies may read the arena from here on (see vr_encode)
    return VR_OK;
}
static void* arena_take(vr_model_s* m, size_t bytes) {
    void* p = m->arena + m->arena_used;
    m->arena_used += (bytes + 63) / 64 * 64;
    return p;
}

static int prof_begin(vr_model_s* m, int cls, hipStream_t s) {
    if (!m->prof_on || (cls >= VR_PROF_DEC_QKV) != (m->prof_level == 2)) return VR_OK;
    auto& p = m->prof[cls];
    if (p.used + 2 > p.ev.size()) {
        for (int i = 0; i < 64; ++i) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); p.ev.push_back(e); }
    }
    HIPCHK(hipEventRecord(p.ev[p.used], s));
    return VR_OK;
}
static int prof_end(vr_model_s* m, int cls, double flops, hipStream_t s) {
    if (!m->prof_on || (cls >= VR_PROF_DEC_QKV) != (m->prof_level == 2)) return VR_OK;
    auto& p = m->prof[cls];
    HIPCHK(hipEventRecord(p.ev[p.used + 1], s));
    p.used += 2; p.launches += 1; p.flops += flops;
    return VR_OK;
}
static int prof_collect(vr_model_s* m) {
    HIPCHK(hipDeviceSynchronize());
    for (auto& p : m->prof) {
        for (size_t i = 0; i + 1 < p.used; i += 2) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]));
            p.ms += ms;
        }
        p.used = 0;
    }
    return VR_OK;
}

// ------------------------------------------------------------------------------ create ---
extern "C" int vr_model_create(int device_id, const vr_config_t* cfg, vr_model_t* out) {
    if (!cfg || !out) return fail(VR_ERR_INVALID, "cfg/out is NULL");
    const vr_config_t& c = *cfg;
    if (c.vit_dim % c.vit_heads || c.vit_dim / c.vit_heads != 72)
        return fail(VR_ERR_INVALID, "ViT head_dim must be 72 (vit_dim %d / heads %d)", c.vit_dim, c.vit_heads);
    if (c.hidden_size % 128) return fail(VR_ERR_INVALID, "hidden_size %d must be a multiple of 128", c.hidden_size);
    if (c.hidden_size % c.num_heads || c.hidden_size / c.num_heads != 64)
        return fail(VR_ERR_INVALID, "decoder head_dim must be 64");
    if (c.intermediate_size % 64) return fail(VR_ERR_INVALID, "intermediate_size must be a multiple of 64");
    if (c.query_num != 64) return fail(VR_ERR_INVALID, "query_num must be 64");
    if (c.vit_dim % 4 || c.hidden_size > 2560 || c.vit_dim > 2560) return fail(VR_ERR_INVALID, "dims out of range");
    if (c.max_images <= 0 || c.max_patches <= 0 || c.max_tokens <= 0 || c.max_seqs <= 0)
        return fail(VR_ERR_INVALID, "workspace limits must be positive");
    VRCHK(set_dev(device_id));
    vr_model_s* m = new vr_model_s();
    m->device = device_id;
    m->c = c;
    m->D = c.vit_dim; m->Dp = pad128(c.vit_dim);
    m->F = c.vit_hidden; m->Fp = pad128(c.vit_hidden);
    m->E = c.hidden_size; m->I = c.intermediate_size; m->Ip = pad128(c.intermediate_size);
    m->Kpe = 3 * c.patch_size * c.patch_size; m->Kpe_p = pad128(m->Kpe);
    m->Q = c.query_num;
    m->blocks.resize(c.vit_depth);
    m->layers.resize(c.num_layers);
    *out = m;
    return VR_OK;
}

extern "C" int vr_model_destroy(vr_model_t m) {
    if (!m) return VR_OK;
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    if (!m->borrowed) {
        auto fl = [](Linear& l) { l.w.free(); l.b.free(); };
        fl(m->patch); fl(m->r_kvproj); fl(m->r_kv); fl(m->r_out); fl(m->r_proj);
        for (auto& b : m->blocks) {
            fl(b.qkv); fl(b.proj); fl(b.fc1); fl(b.fc2); b.n1w.v.free(); b.n1b.v.free(); b.n2w.v.free(); b.n2b.v.free();
        }
        for (auto& l : m->layers) { fl(l.qkv); fl(l.o); fl(l.gu); fl(l.down); fl(l.qkv_lo); fl(l.o_lo); fl(l.gu_lo); fl(l.down_lo); l.ln1.v.free(); l.ln2.v.free(); }
        for (Vec* v : {&m->vit_nw, &m->vit_nb, &m->r_lnq_w, &m->r_lnq_b, &m->r_lnkv_w, &m->r_lnkv_b, &m->r_lnpost_w, &m->r_lnpost_b, &m->final_norm}) v->v.free();
        for (DevBuf* b : {&m->r_q, &m->embed, &m->embed_lo, &m->rope}) b->free();
    }
    for (auto& g : m->grids) { g.second.vit_pos.free(); g.second.pos_k.free(); }
    for (auto& pc : m->prof) for (hipEvent_t e : pc.ev) (void)hipEventDestroy(e);
    if (m->arena) (void)hipHostFree(m->arena);
    if (m->arena_ev) (void)hipEventDestroy(m->arena_ev);
    for (DevBuf* b : {&m->w_hvit, &m->w_xn, &m->w_qkv, &m->w_att, &m->w_mlp,
                      &m->w_kv32, &m->w_xkv, &m->w_KV, &m->w_ratt, &m->w_rout, &m->w_rln, &m->w_h, &m->w_dxn, &m->w_part, &m->w_dqkv,
                      &m->w_datt, &m->w_dact, &m->w_cu, &m->w_ids, &m->w_seq, &m->w_pos, &m->w_rowmap, &m->w_imgptr,
                      &m->w_pix, &m->w_out, &m->w_hp_hi, &m->w_hp_planes, &m->w_hp_qkv, &m->w_hp_att, &m->w_hp_gu, &m->w_seqof, &m->w_hp_part, &m->w_hidden})
        b->free();
    delete m;
    return VR_OK;
}

extern "C" int vr_model_load_weight(vr_model_t m, const char* name_c, const void* data, const int64_t* shape,
                                    int32_t ndim, int32_t dtype, int32_t on_device) {
    if (!m || !name_c || !data || !shape) return fail(VR_ERR_INVALID, "NULL argument");
    if (dtype != VR_DTYPE_F32 && dtype != VR_DTYPE_BF16) return fail(VR_ERR_INVALID, "bad dtype %d", dtype);
    VRCHK(set_dev(m->device));
    const std::string name(name_c);
    const int bf = dtype == VR_DTYPE_BF16;
    size_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
    const vr_config_t& c = m->c;
    const int D = m->D, F = m->F, E = m->E, I = m->I;
    auto bad_shape = [&]() { return fail(VR_ERR_INVALID, "unexpected shape for %s", name_c); };

    // keys the embedding path does not use
    if (name.rfind("llm.lm_head.", 0) == 0 || name.rfind("vpm.attn_pool.", 0) == 0 ||
        name == "resampler.pos_embed" || name.find("rotary_emb") != std::string::npos)
        return VR_OK;

    Staged st;
    VRCHK(stage(data, numel * (bf ? 2 : 4), on_device, st));
    const void* src = st.dev;
    m->finalized = false;

    if (name == "vpm.patch_embed.proj.weight") {
        if (!shape_is(shape, ndim, {D, 3, c.patch_size, c.patch_size})) return bad_shape();
        Linear& L = m->patch;      // columns permuted to the image's byte order inside a patch (patch_embed.hip)
        if (!L.w.p) { L.n = D; L.k = m->Kpe; L.n_pad = pad128(D); L.k_pad = pad128(m->Kpe); VRCHK(L.w.alloc((size_t)pad256(D) * L.k_pad * 2)); }
        HIPCHK(launch_pack_patch_weight(src, bf, D, c.patch_size, L.w.p, L.k_pad, 0));
        HIPCHK(hipDeviceSynchronize());
        L.has_w = true;
        return VR_OK;
    }
    if (name == "vpm.patch_embed.proj.bias") { if (numel != (size_t)D) return bad_shape(); return load_bias_part(m->patch, D, src, bf, D, 0); }
    if (name == "vpm.pos_embed") {
        if (numel != (size_t)c.vit_pos_grid * c.vit_pos_grid * D) return bad_shape();
        m->has_pos = true;
        m->grids.clear();
        return to_host_f32(src, bf, numel, m->pos_embed_host);
    }
    if (name == "vpm.norm.weight") { if (numel != (size_t)D) return bad_shape(); return load_vec(m->vit_nw, src, bf, D, m->Dp); }
    if (name == "vpm.norm.bias") { if (numel != (size_t)D) return bad_shape(); return load_vec(m->vit_nb, src, bf, D, m->Dp); }
    if (name.rfind("vpm.blocks.", 0) == 0) {
        int n = -1, off = 0;
        if (sscanf(name.c_str(), "vpm.blocks.%d.%n", &n, &off) < 1) return fail(VR_ERR_INVALID, "bad key %s", name_c);
        if (n >= c.vit_depth) return VR_OK;     // dropped last block (modeling_minicpmv.py:70-71)
        const std::string sub = name.substr(off);
        VitBlock& b = m->blocks[n];
        if (sub == "norm1.weight") { if (numel != (size_t)D) return bad_shape(); return load_vec(b.n1w, src, bf, D, m->Dp); }
        if (sub == "norm1.bias") { if (numel != (size_t)D) return bad_shape(); return load_vec(b.n1b, src, bf, D, m->Dp); }
        if (sub == "norm2.weight") { if (numel != (size_t)D) return bad_shape(); return load_vec(b.n2w, src, bf, D, m->Dp); }
        if (sub == "norm2.bias") { if (numel != (size_t)D) return bad_shape(); return load_vec(b.n2b, src, bf, D, m->Dp); }
        if (sub == "attn.qkv.weight") { if (!shape_is(shape, ndim, {3 * D, D})) return bad_shape(); return load_linear_part(b.qkv, 3 * D, D, src, bf, 3 * D, D, 0, 3 * D, 0, 0); }
        if (sub == "attn.qkv.bias") { if (numel != (size_t)3 * D) return bad_shape(); return load_bias_part(b.qkv, 3 * D, src, bf, 3 * D, 0); }
        if (sub == "attn.proj.weight") { if (!shape_is(shape, ndim, {D, D})) return bad_shape(); return load_linear_part(b.proj, D, D, src, bf, D, D, 0, D, 0, 0); }
        if (sub == "attn.proj.bias") { if (numel != (size_t)D) return bad_shape(); return load_bias_part(b.proj, D, src, bf, D, 0); }
        if (sub == "mlp.fc1.weight") { if (!shape_is(shape, ndim, {F, D})) return bad_shape(); return load_linear_part(b.fc1, F, D, src, bf, F, D, 0, F, 0, 0); }
        if (sub == "mlp.fc1.bias") { if (numel != (size_t)F) return bad_shape(); return load_bias_part(b.fc1, F, src, bf, F, 0); }
        if (sub == "mlp.fc2.weight") { if (!shape_is(shape, ndim, {D, F})) return bad_shape(); return load_linear_part(b.fc2, D, F, src, bf, D, F, 0, D, 0, 0); }
        if (sub == "mlp.fc2.bias") { if (numel != (size_t)D) return bad_shape(); return load_bias_part(b.fc2, D, src, bf, D, 0); }
        return fail(VR_ERR_INVALID, "unknown ViT key %s", name_c);
    }
    if (name.rfind("resampler.", 0) == 0) {
        const std::string sub = name.substr(10);
        if (sub == "query") { if (!shape_is(shape, ndim, {m->Q, E})) return bad_shape(); m->has_query = true; return to_host_f32(src, bf, numel, m->r_query_host); }
        if (sub == "kv_proj.weight") { if (!shape_is(shape, ndim, {E, D})) return bad_shape(); return load_linear_part(m->r_kvproj, E, D, src, bf, E, D, 0, E, 0, 0); }
        if (sub == "attn.in_proj_weight") {
            if (!shape_is(shape, ndim, {3 * E, E})) return bad_shape();
            std::vector<float> all;
            VRCHK(to_host_f32(src, bf, (size_t)E * E, all));      // q rows only
            m->r_wq_host.swap(all);
            m->has_inproj = true;
            const char* kv_src = (const char*)src + (size_t)E * E * (bf ? 2 : 4);
            return load_linear_part(m->r_kv, 2 * E, E, kv_src, bf, 2 * E, E, 0, 2 * E, 0, 0);
        }
        if (sub == "attn.in_proj_bias") {
            if (numel != (size_t)3 * E) return bad_shape();
            VRCHK(to_host_f32(src, bf, (size_t)E, m->r_bq_host));
            m->has_inproj_b = true;
            const char* kv_src = (const char*)src + (size_t)E * (bf ? 2 : 4);
            return load_bias_part(m->r_kv, 2 * E, kv_src, bf, 2 * E, 0);
        }
        if (sub == "attn.out_proj.weight") { if (!shape_is(shape, ndim, {E, E})) return bad_shape(); return load_linear_part(m->r_out, E, E, src, bf, E, E, 0, E, 0, 0); }
        if (sub == "attn.out_proj.bias") { if (numel != (size_t)E) return bad_shape(); return load_bias_part(m->r_out, E, src, bf, E, 0); }
        if (sub == "proj") { if (!shape_is(shape, ndim, {E, E})) return bad_shape(); return load_linear_part(m->r_proj, E, E, src, bf, E, E, 1, E, 0, 0); }
        Vec* v = nullptr;
        if (sub == "ln_q.weight") v = &m->r_lnq_w; else if (sub == "ln_q.bias") v = &m->r_lnq_b;
        else if (sub == "ln_kv.weight") v = &m->r_lnkv_w; else if (sub == "ln_kv.bias") v = &m->r_lnkv_b;
        else if (sub == "ln_post.weight") v = &m->r_lnpost_w; else if (sub == "ln_post.bias") v = &m->r_lnpost_b;
        if (!v) return fail(VR_ERR_INVALID, "unknown resampler key %s", name_c);
        if (numel != (size_t)E) return bad_shape();
        return load_vec(*v, src, bf, E, E);
    }
    if (name == "llm.model.embed_tokens.weight") {
        if (!shape_is(shape, ndim, {c.vocab_size, E})) return bad_shape();
        VRCHK(m->embed.alloc(numel * 2));
        HIPCHK(launch_pack_weight(src, bf, c.vocab_size, E, E, 0, m->embed.p, E, c.vocab_size, 0, 0, 0));
        m->has_embed_lo = false;
        if (!bf && c.text_split_precision) {
            VRCHK(m->embed_lo.alloc(numel * 2));
            HIPCHK(launch_pack_weight(src, bf, c.vocab_size, E, E, 0, m->embed_lo.p, E, c.vocab_size, 0, 0, 0, 1));
            m->has_embed_lo = true;
        }
        HIPCHK(hipDeviceSynchronize());
        m->has_embed = true;
        return VR_OK;
    }
    if (name == "llm.model.norm.weight") { if (numel != (size_t)E) return bad_shape(); return load_vec(m->final_norm, src, bf, E, E); }
    if (name.rfind("llm.model.layers.", 0) == 0) {
        int n = -1, off = 0;
        if (sscanf(name.c_str(), "llm.model.layers.%d.%n", &n, &off) < 1) return fail(VR_ERR_INVALID, "bad key %s", name_c);
        if (n >= c.num_layers) return fail(VR_ERR_INVALID, "layer index %d out of range", n);
        const std::string sub = name.substr(off);
        DecLayer& l = m->layers[n];
        if (sub == "input_layernorm.weight") { if (numel != (size_t)E) return bad_shape(); return load_vec(l.ln1, src, bf, E, E); }
        if (sub == "post_attention_layernorm.weight") { if (numel != (size_t)E) return bad_shape(); return load_vec(l.ln2, src, bf, E, E); }
        // fp32 source weights also leave their low halves (w - bf16(w)) for the split-precision text path; a bf16
        // checkpoint has none (the two activation halves against the one weight are then the whole product)
        const bool lo = !bf && c.text_split_precision;
        for (int part = 0; part < 3; ++part) {
            static const char* nm[3] = {"self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight"};
            if (sub == nm[part]) {
                if (!shape_is(shape, ndim, {E, E})) return bad_shape();
                l.parts_qkv |= 1 << part;
                if (lo) VRCHK(load_linear_part(l.qkv_lo, 3 * E, E, src, bf, E, E, 0, E, 0, part * E, 1));
                return load_linear_part(l.qkv, 3 * E, E, src, bf, E, E, 0, E, 0, part * E);
            }
        }
        if (sub == "self_attn.o_proj.weight") {
            if (!shape_is(shape, ndim, {E, E})) return bad_shape();
            if (lo) VRCHK(load_linear_part(l.o_lo, E, E, src, bf, E, E, 0, E, 0, 0, 1));
            return load_linear_part(l.o, E, E, src, bf, E, E, 0, E, 0, 0);
        }
        if (sub == "mlp.gate_proj.weight" || sub == "mlp.up_proj.weight") {
            if (!shape_is(shape, ndim, {I, E})) return bad_shape();
            const int up = sub == "mlp.up_proj.weight";
            l.parts_gu |= 1 << up;
            // 16-row interleave: [16 gate | 16 up | ...] (EPI_SWIGLU)
            if (lo) VRCHK(load_linear_part(l.gu_lo, 2 * I, E, src, bf, I, E, 0, 16, 32, up * 16, 1));
            return load_linear_part(l.gu, 2 * I, E, src, bf, I, E, 0, 16, 32, up * 16);
        }
        if (sub == "mlp.down_proj.weight") {
            if (!shape_is(shape, ndim, {E, I})) return bad_shape();
            if (lo) VRCHK(load_linear_part(l.down_lo, E, I, src, bf, E, I, 0, E, 0, 0, 1));
            return load_linear_part(l.down, E, I, src, bf, E, I, 0, E, 0, 0);
        }
        return fail(VR_ERR_INVALID, "unknown decoder key %s", name_c);
    }
    return fail(VR_ERR_INVALID, "unknown weight key %s", name_c);
}

// ---------------------------------------------------------------------- derived tables ---
// fp32 sincos table of resampler.py:38-90 (numpy float32 arithmetic restated)
static void sincos_2d_host(int E, int gh, int gw, std::vector<float>& out) {
    const int half = E / 2, quarter = half / 2;
    out.assign((size_t)gh * gw * E, 0.f);
    std::vector<float> omega(quarter);
    for (int i = 0; i < quarter; ++i) {
        float o = (float)i / ((float)half / 2.0f);
        omega[i] = 1.0f / powf(10000.0f, o);
    }
    for (int y = 0; y < gh; ++y)
        for (int x = 0; x < gw; ++x) {
            float* row = out.data() + ((size_t)y * gw + x) * E;
            // first half <- grid[0] = column index (meshgrid(w, h), "w goes first"); second <- row index
            for (int i = 0; i < quarter; ++i) {
                const float a = (float)x * omega[i], b = (float)y * omega[i];
                row[i] = sinf(a); row[quarter + i] = cosf(a);
                row[half + i] = sinf(b); row[half + quarter + i] = cosf(b);
            }
        }
}

// bicubic (a = -0.5) anti-aliased separable resample, align_corners=False: the algorithm of
// F.interpolate(mode="bicubic", antialias=True) used by timm's resample_abs_pos_embed
// (timm/layers/pos_embed.py:46).  in [gi][gi][D] -> out [gh][gw][D].
static inline float cubic_aa(float x) {
    const float a = -0.5f;
    x = fabsf(x);
    if (x < 1.0f) return ((a + 2.0f) * x - (a + 3.0f)) * x * x + 1.0f;
    if (x < 2.0f) return (((x - 5.0f) * x + 8.0f) * x - 4.0f) * a;
    return 0.0f;
}
static void aa_weights(int in, int out, std::vector<int>& xmin, std::vector<int>& xsize, std::vector<float>& w, int& maxk) {
    const float scale = (float)in / (float)out;
    const float support = (scale >= 1.0f) ? 2.0f * scale : 2.0f;
    const float invscale = (scale >= 1.0f) ? 1.0f / scale : 1.0f;
    maxk = (int)ceilf(support) * 2 + 1;
    xmin.resize(out); xsize.resize(out); w.assign((size_t)out * maxk, 0.f);
    for (int i = 0; i < out; ++i) {
        const float center = scale * ((float)i + 0.5f);
        int lo = std::max(0, (int)(center - support + 0.5f));
        int hi = std::min(in, (int)(center + support + 0.5f));
        xmin[i] = lo; xsize[i] = hi - lo;
        float tot = 0.f;
        for (int j = 0; j < xsize[i]; ++j) {
            const float ww = cubic_aa(((float)(j + lo) - center + 0.5f) * invscale);
            w[(size_t)i * maxk + j] = ww; tot += ww;
        }
        for (int j = 0; j < xsize[i]; ++j) w[(size_t)i * maxk + j] /= tot;
    }
}
static void resample_pos_host(const std::vector<float>& pe, int gi, int D, int gh, int gw, std::vector<float>& out) {
    out.assign((size_t)gh * gw * D, 0.f);
    if (gh == gi && gw == gi) { out = pe; return; }
    std::vector<int> xm, xs, ym, ys; std::vector<float> xw, yw; int xk, yk;
    aa_weights(gi, gw, xm, xs, xw, xk);
    aa_weights(gi, gh, ym, ys, yw, yk);
    std::vector<float> tmp((size_t)gi * gw * D, 0.f);      // horizontal pass
    for (int y = 0; y < gi; ++y)
        for (int x = 0; x < gw; ++x) {
            float* o = tmp.data() + ((size_t)y * gw + x) * D;
            for (int j = 0; j < xs[x]; ++j) {
                const float ww = xw[(size_t)x * xk + j];
                const float* s = pe.data() + ((size_t)y * gi + xm[x] + j) * D;
                for (int d = 0; d < D; ++d) o[d] += ww * s[d];
            }
        }
    for (int y = 0; y < gh; ++y)                             // vertical pass
        for (int x = 0; x < gw; ++x) {
            float* o = out.data() + ((size_t)y * gw + x) * D;
            for (int j = 0; j < ys[y]; ++j) {
                const float ww = yw[(size_t)y * yk + j];
                const float* s = tmp.data() + ((size_t)(ym[y] + j) * gw + x) * D;
                for (int d = 0; d < D; ++d) o[d] += ww * s[d];
            }
        }
}

constexpr int DEC_KSPLIT_MAX = 3;   // decoder o / down projections: split-K factor when the tile grid is small

static GemmArgs gemm_args(const void* A, int lda, const Linear& L, int M, void* out, int ldo) {
    GemmArgs a{};
    a.A = A; a.lda = lda; a.W = L.w.p; a.ldw = L.k_pad; a.M = M; a.N = L.n_pad; a.K = L.k_pad;
    a.bias = L.has_b ? L.b.as<float>() : nullptr;
    a.out = out; a.ldo = ldo; a.alpha = 1.0f;
    return a;
}

static int alloc_workspace(vr_model_s* m) {
    const vr_config_t& c = m->c;
    const int64_t M = pad256l((int64_t)c.max_images * c.max_patches);
    const int64_t T = pad256l(c.max_tokens);
    const int64_t R = pad256l((int64_t)c.max_images * m->Q);
    m->Mcap = M; m->Tcap = T; m->Rcap = R;
    const int E = m->E, Dp = m->Dp;
    VRCHK(m->w_hvit.alloc((size_t)M * Dp * 4));
    VRCHK(m->w_xn.alloc((size_t)M * Dp * 2));
    VRCHK(m->w_qkv.alloc((size_t)M * pad128(3 * m->D) * 2));
    VRCHK(m->w_att.alloc((size_t)M * Dp * 2));
    VRCHK(m->w_mlp.alloc((size_t)M * m->Fp * 2));
    VRCHK(m->w_kv32.alloc((size_t)M * E * 4));
    VRCHK(m->w_xkv.alloc((size_t)M * E * 2));
    VRCHK(m->w_KV.alloc((size_t)M * 2 * E * 2));
    VRCHK(m->w_ratt.alloc((size_t)R * E * 2));
    VRCHK(m->w_rout.alloc((size_t)R * E * 4));
    VRCHK(m->w_rln.alloc((size_t)R * E * 2));
    VRCHK(m->w_h.alloc((size_t)T * E * 4));
    VRCHK(m->w_dxn.alloc((size_t)T * E * 2));
    VRCHK(m->w_part.alloc((size_t)DEC_KSPLIT_MAX * T * E * 4));
    VRCHK(m->w_dqkv.alloc((size_t)T * 3 * E * 2));
    VRCHK(m->w_datt.alloc((size_t)T * E * 2));
    VRCHK(m->w_dact.alloc((size_t)T * m->Ip * 2));
    VRCHK(m->w_cu.alloc((size_t)(c.max_images + c.max_seqs + 8) * 2 * 4));
    VRCHK(m->w_ids.alloc((size_t)T * 4));
    VRCHK(m->w_seq.alloc((size_t)(c.max_seqs + 1) * 4));
    VRCHK(m->w_pos.alloc((size_t)T * 4));
    VRCHK(m->w_rowmap.alloc((size_t)R * 4));
    VRCHK(m->w_imgptr.alloc((size_t)c.max_images * 8));
    VRCHK(m->w_out.alloc((size_t)c.max_seqs * E * 4));
    if (c.text_split_precision) {
        const int Kmax = std::max(E, m->Ip);
        VRCHK(m->w_hp_hi.alloc((size_t)2 * T * Kmax * 2));       // [hi rows | lo rows]: the lo half starts right behind the batch's T hi rows
        VRCHK(m->w_hp_qkv.alloc((size_t)T * 3 * E * 4));
        VRCHK(m->w_hp_att.alloc((size_t)T * E * 4));
        VRCHK(m->w_hp_gu.alloc((size_t)T * pad128(2 * m->I) * 4));
        VRCHK(m->w_seqof.alloc((size_t)T * 4));
        // split-K planes of the weight-streaming path of short batches: 3 passes x 32 rows x (ksplit * n_pad <= 256 tiles of 256)
        VRCHK(m->w_hp_part.alloc((size_t)3 * 32 * 65536 * 4));
    }
    return VR_OK;
}

extern "C" int vr_model_finalize(vr_model_t m) {
    if (!m) return fail(VR_ERR_INVALID, "NULL model");
    VRCHK(set_dev(m->device));
    const vr_config_t& c = m->c;
    const int E = m->E;
    // ---- completeness
    auto need = [&](bool ok, const char* what) { return ok ? VR_OK : fail(VR_ERR_STATE, "missing weight: %s", what); };
    VRCHK(need(m->patch.has_w && m->patch.has_b, "vpm.patch_embed.proj"));
    VRCHK(need(m->has_pos, "vpm.pos_embed"));
    VRCHK(need(m->vit_nw.ok && m->vit_nb.ok, "vpm.norm"));
    for (int n = 0; n < c.vit_depth; ++n) {
        const VitBlock& b = m->blocks[n];
        const bool ok = b.n1w.ok && b.n1b.ok && b.n2w.ok && b.n2b.ok && b.qkv.has_w && b.qkv.has_b && b.proj.has_w &&
                        b.proj.has_b && b.fc1.has_w && b.fc1.has_b && b.fc2.has_w && b.fc2.has_b;
        if (!ok) return fail(VR_ERR_STATE, "missing weight in vpm.blocks.%d", n);
    }
    VRCHK(need(m->has_query && m->has_inproj && m->has_inproj_b, "resampler.query / attn.in_proj"));
    VRCHK(need(m->r_kvproj.has_w && m->r_kv.has_w && m->r_kv.has_b && m->r_out.has_w && m->r_out.has_b && m->r_proj.has_w,
               "resampler linear weights"));
    VRCHK(need(m->r_lnq_w.ok && m->r_lnq_b.ok && m->r_lnkv_w.ok && m->r_lnkv_b.ok && m->r_lnpost_w.ok && m->r_lnpost_b.ok,
               "resampler layer norms"));
    VRCHK(need(m->has_embed && m->final_norm.ok, "llm.model.embed_tokens / norm"));
    for (int n = 0; n < c.num_layers; ++n) {
        const DecLayer& l = m->layers[n];
        const bool ok = l.ln1.ok && l.ln2.ok && l.parts_qkv == 7 && l.parts_gu == 3 && l.o.has_w && l.down.has_w;
        if (!ok) return fail(VR_ERR_STATE, "missing weight in llm.model.layers.%d", n);
    }
    // ---- resampler query projection (batch-invariant, computed once in fp32 on the host):
    //      q = (ln_q(query) + sincos(8x8)) @ Wq^T + bq          resampler.py:157-160
    {
        const int Q = m->Q;
        std::vector<float> lw(E), lb(E), pq;
        HIPCHK(hipMemcpy(lw.data(), m->r_lnq_w.v.p, (size_t)E * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(lb.data(), m->r_lnq_b.v.p, (size_t)E * 4, hipMemcpyDeviceToHost));
        const int g = (int)lround(sqrt((double)Q));
        sincos_2d_host(E, g, g, pq);
        std::vector<float> x((size_t)Q * E), out((size_t)Q * E);
        for (int q = 0; q < Q; ++q) {
            const float* s = m->r_query_host.data() + (size_t)q * E;
            double mu = 0; for (int i = 0; i < E; ++i) mu += s[i]; mu /= E;
            double var = 0; for (int i = 0; i < E; ++i) { const double d = s[i] - mu; var += d * d; } var /= E;
            const float rstd = (float)(1.0 / sqrt(var + (double)c.resampler_ln_eps));
            for (int i = 0; i < E; ++i) x[(size_t)q * E + i] = ((float)(s[i] - mu)) * rstd * lw[i] + lb[i] + pq[(size_t)q * E + i];
        }
        for (int q = 0; q < Q; ++q)
            for (int n = 0; n < E; ++n) {
                const float* wr = m->r_wq_host.data() + (size_t)n * E;
                const float* xr = x.data() + (size_t)q * E;
                float acc = 0.f;
                for (int i = 0; i < E; ++i) acc += xr[i] * wr[i];
                out[(size_t)q * E + n] = acc + m->r_bq_host[n];
            }
        DevBuf t;
        VRCHK(t.alloc(out.size() * 4));
        HIPCHK(hipMemcpy(t.p, out.data(), out.size() * 4, hipMemcpyHostToDevice));
        VRCHK(m->r_q.alloc((size_t)pad128(Q) * E * 2));
        HIPCHK(launch_f32_to_bf16(t.as<float>(), m->r_q.p, out.size(), 0));
        HIPCHK(hipDeviceSynchronize());
        t.free();
    }
    // ---- RoPE table [pos][cos 32 | sin 32], fp32 (modeling_minicpm.py:142-172)
    {
        m->rope_len = std::max(c.max_tokens, 16);
        std::vector<float> tab((size_t)m->rope_len * 64);
        float inv[32];
        for (int i = 0; i < 32; ++i) inv[i] = 1.0f / powf(c.rope_theta, (float)(2 * i) / 64.0f);
        for (int p = 0; p < m->rope_len; ++p)
            for (int i = 0; i < 32; ++i) {
                const float a = (float)p * inv[i];
                tab[(size_t)p * 64 + i] = cosf(a);
                tab[(size_t)p * 64 + 32 + i] = sinf(a);
            }
        VRCHK(m->rope.alloc(tab.size() * 4));
        HIPCHK(hipMemcpy(m->rope.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    }
    // ---- split-precision text path: a checkpoint whose fp32 weights are bf16-exact (or that came as bf16) has no
    //      low halves — drop the all-zero buffers, the passes over them would add nothing but weight reads
    if (c.text_split_precision) {
        DevBuf flag;
        VRCHK(flag.alloc(4));
        for (auto& l : m->layers)
            for (Linear* L : {&l.qkv_lo, &l.o_lo, &l.gu_lo, &l.down_lo})
                if (L->has_w) HIPCHK(launch_any_nonzero16(L->w.p, L->w.bytes / 2, flag.as<int>(), 0));
        int any = 0;
        HIPCHK(hipMemcpy(&any, flag.p, 4, hipMemcpyDeviceToHost));
        if (!any)
            for (auto& l : m->layers)
                for (Linear* L : {&l.qkv_lo, &l.o_lo, &l.gu_lo, &l.down_lo}) { L->w.free(); L->has_w = false; }
        if (m->has_embed_lo) {
            HIPCHK(hipMemset(flag.p, 0, 4));
            HIPCHK(launch_any_nonzero16(m->embed_lo.p, m->embed_lo.bytes / 2, flag.as<int>(), 0));
            HIPCHK(hipMemcpy(&any, flag.p, 4, hipMemcpyDeviceToHost));
            if (!any) { m->embed_lo.free(); m->has_embed_lo = false; }
        }
    }
    if (!m->w_h.p) VRCHK(alloc_workspace(m));
    m->finalized = true;
    return VR_OK;
}

// per-grid constants: resampled ViT pos-embed and the k-side position term of the resampler
static int get_grid(vr_model_s* m, int gh, int gw, GridTables** out) {
    auto key = std::make_pair(gh, gw);
    auto it = m->grids.find(key);
    if (it != m->grids.end()) { *out = &it->second; return VR_OK; }
    const int N = gh * gw, D = m->D, Dp = m->Dp, E = m->E;
    GridTables g;
    g.gh = gh; g.gw = gw;
    {   // K3: timm resample_abs_pos_embed, once per grid instead of once per forward
        std::vector<float> rs, padded((size_t)N * Dp, 0.f);
        resample_pos_host(m->pos_embed_host, m->c.vit_pos_grid, D, gh, gw, rs);
        for (int r = 0; r < N; ++r) memcpy(padded.data() + (size_t)r * Dp, rs.data() + (size_t)r * D, (size_t)D * 4);
        VRCHK(g.vit_pos.alloc(padded.size() * 4));
        HIPCHK(hipMemcpy(g.vit_pos.p, padded.data(), padded.size() * 4, hipMemcpyHostToDevice));
    }
    {   // K10/K11: k = (x + pos) Wk^T + bk = x Wk^T + bk + (pos Wk^T); the last term is a
        // per-position bias computed once per grid with two bf16 passes (hi + lo split of the
        // fp32 sincos table keeps ~16 mantissa bits).
        std::vector<float> sc;
        sincos_2d_host(E, gh, gw, sc);
        const int Np = pad128(N);
        DevBuf f32, hi, lo, tmp;
        VRCHK(f32.alloc((size_t)Np * E * 4));
        VRCHK(hi.alloc((size_t)Np * E * 2));
        VRCHK(lo.alloc((size_t)Np * E * 2));
        VRCHK(tmp.alloc((size_t)Np * E * 4));
        VRCHK(g.pos_k.alloc((size_t)Np * E * 4));
        HIPCHK(hipMemcpy(f32.p, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(launch_split_bf16(f32.as<float>(), hi.p, lo.p, (size_t)N * E, 0));
        GemmArgs a{};
        a.A = hi.p; a.lda = E; a.W = m->r_kv.w.p; a.ldw = m->r_kv.k_pad; a.M = N; a.N = E; a.K = E;
        a.out = tmp.p; a.ldo = E; a.alpha = 1.f;
        HIPCHK(launch_gemm(a, EPI_F32, GEMM_VARIANT_GLDS, 0));
        a.A = lo.p; a.resid = tmp.as<float>(); a.out = g.pos_k.p;
        HIPCHK(launch_gemm(a, EPI_RESID, GEMM_VARIANT_GLDS, 0));
        HIPCHK(hipDeviceSynchronize());
        f32.free(); hi.free(); lo.free(); tmp.free();
    }
    auto ins = m->grids.emplace(key, std::move(g));
    *out = &ins.first->second;
    return VR_OK;
}

// ---------------------------------------------------------------------------------- taps ---
static int tap_store(vr_model_s* m, const char* name, const void* dev, int64_t rows, int64_t cols, int64_t ld,
                     bool is_bf16, hipStream_t s) {
    if (!m->taps_on) return VR_OK;
    HIPCHK(hipStreamSynchronize(s));
    Tap& t = m->taps[name];
    t.rows = rows; t.cols = cols;
    t.data.resize((size_t)rows * cols);
    const size_t esz = is_bf16 ? 2 : 4;
    std::vector<char> raw((size_t)rows * ld * esz);
    HIPCHK(hipMemcpy(raw.data(), dev, raw.size(), hipMemcpyDeviceToHost));
    for (int64_t r = 0; r < rows; ++r)
        for (int64_t cc = 0; cc < cols; ++cc) {
            if (is_bf16) {
                uint16_t b; memcpy(&b, raw.data() + ((size_t)r * ld + cc) * 2, 2);
                uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4);
                t.data[(size_t)r * cols + cc] = f;
            } else {
                float f; memcpy(&f, raw.data() + ((size_t)r * ld + cc) * 4, 4);
                t.data[(size_t)r * cols + cc] = f;
            }
        }
    return VR_OK;
}

extern "C" int vr_model_set_pooling(vr_model_t m, int32_t mode) {
    if (!m) return fail(VR_ERR_INVALID, "NULL model");
    if (mode < VR_POOL_WMEAN || mode > VR_POOL_CLS) return fail(VR_ERR_INVALID, "pooling mode %d: 0 wmean, 1 mean, 2 lasttoken, 3 cls", mode);
    m->pool_mode = mode;
    return VR_OK;
}

extern "C" int vr_model_set_taps(vr_model_t m, int32_t enable) {
    if (!m) return fail(VR_ERR_INVALID, "NULL model");
    m->taps_on = enable != 0;
    if (!enable) m->taps.clear();
    return VR_OK;
}

extern "C" int vr_model_tap(vr_model_t m, const char* name, float* out, int64_t rows, int64_t cols) {
    if (!m || !name || !out) return fail(VR_ERR_INVALID, "NULL argument");
    auto it = m->taps.find(name);
    if (it == m->taps.end()) return fail(VR_ERR_STATE, "tap %s not recorded (enable taps, then encode)", name);
    const Tap& t = it->second;
    if (rows > t.rows || cols != t.cols) return fail(VR_ERR_INVALID, "tap %s is [%lld][%lld]", name, (long long)t.rows, (long long)t.cols);
    memcpy(out, t.data.data(), (size_t)rows * cols * 4);
    return VR_OK;
}

// -------------------------------------------------------------------------------- encode ---
// ViT + resampler for `n` same-shape slices; writes resampler rows into the decoder stream.
static int run_vision_group(vr_model_s* m, const uint8_t* const* dev_imgs_hostarray, int n, int H, int W,
                            const int32_t* rowmap_host /* [n*Q] */, bool first_group, hipStream_t s) {
    const vr_config_t& c = m->c;
    const int P = c.patch_size, gh = H / P, gw = W / P, N = gh * gw;
    const int D = m->D, Dp = m->Dp, E = m->E, Q = m->Q;
    const int M = n * N;
    GridTables* g = nullptr;
    VRCHK(get_grid(m, gh, gw, &g));

    // dev_imgs_hostarray / rowmap_host live in the pinned arena (see vr_encode)
    HIPCHK(hipMemcpyAsync(m->w_imgptr.p, dev_imgs_hostarray, (size_t)n * 8, hipMemcpyHostToDevice, s));
    int* cu = (int*)arena_take(m, (size_t)2 * (n + 1) * 4);
    for (int i = 0; i <= n; ++i) { cu[i] = i * N; cu[n + 1 + i] = i * Q; }
    HIPCHK(hipMemcpyAsync(m->w_cu.p, cu, (size_t)2 * (n + 1) * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->w_rowmap.p, rowmap_host, (size_t)n * Q * 4, hipMemcpyHostToDevice, s));
    const int* cu_tok = m->w_cu.as<int>();
    const int* cu_qry = m->w_cu.as<int>() + (n + 1);

    float* h = m->w_hvit.as<float>();
    // K1+K2+K3: normalise + patch tiles built in LDS + conv-as-GEMM + bias + resampled pos-embed -> fp32 residual stream
    {
        GemmArgs a = gemm_args(nullptr, 0, m->patch, M, h, Dp);
        a.rowbias = g->vit_pos.as<float>(); a.rowbias_period = N; a.rowbias_ld = Dp; a.rowbias_cols = Dp;
        HIPCHK(launch_patch_embed((const uint8_t* const*)m->w_imgptr.p, n, H, W, P, a, m->Kpe, s));
    }
    if (first_group) VRCHK(tap_store(m, "vit_embed", h, N, D, Dp, false, s));
    const int ldqkv = pad128(3 * D);
    // (the plain bf16 epilogue scales whole 64-column blocks: the q section must end on one, as 16 x 72 does)
    // NOTE for any other reader of w_qkv: its q columns hold q * head_dim^-0.5 * log2(e), not q (AttnArgs::q_prescaled)
    const int vit_hd = c.vit_dim / c.vit_heads;                          // (72: checked at vr_model_create)
    const int vit_qscale_n = D % 64 == 0 ? D : 0;
    const float vit_qscale = (1.0f / sqrtf((float)vit_hd)) * 1.44269504088896340736f;
    for (int l = 0; l < c.vit_depth; ++l) {
        const VitBlock& b = m->blocks[l];
        HIPCHK(launch_layernorm(h, M, D, Dp, b.n1w.v.as<float>(), b.n1b.v.as<float>(), c.vit_ln_eps, m->w_xn.p, Dp, s));
        VRCHK(prof_begin(m, VR_PROF_VIT_QKV, s));
        {
            // the q columns leave with head_dim^-0.5 * log2(e) folded in before their one bf16 rounding (attention_w.hip starts its
            // score accumulators at -m and feeds them to v_exp_f32 as they come out)
            GemmArgs a = gemm_args(m->w_xn.p, Dp, b.qkv, M, m->w_qkv.p, ldqkv);
            a.col_scale = vit_qscale; a.col_scale_n = vit_qscale_n;
            HIPCHK(launch_gemm(a, EPI_BF16, GEMM_VARIANT_AUTO, s));
        }
        VRCHK(prof_end(m, VR_PROF_VIT_QKV, 2.0 * M * D * 3 * D, s));
        {
            AttnArgs a{};
            a.q = m->w_qkv.p; a.ldq = ldqkv;
            a.k = (const char*)m->w_qkv.p + (size_t)D * 2; a.ldk = ldqkv;
            a.v = (const char*)m->w_qkv.p + (size_t)2 * D * 2; a.ldv = ldqkv;
            a.out = m->w_att.p; a.ldo = Dp; a.cu_q = cu_tok; a.cu_kv = cu_tok; a.B = n; a.heads = c.vit_heads;
            a.head_dim = vit_hd; a.max_q = N; a.causal = 0; a.q_shared = 0; a.scale = 1.0f / sqrtf((float)vit_hd); a.q_prescaled = vit_qscale_n > 0;
            VRCHK(prof_begin(m, VR_PROF_VIT_ATTN, s));
            HIPCHK(launch_attention(a, s));
            VRCHK(prof_end(m, VR_PROF_VIT_ATTN, 4.0 * n * (double)N * N * D, s));
        }
        VRCHK(prof_begin(m, VR_PROF_VIT_PROJ, s));
        {
            GemmArgs a = gemm_args(m->w_att.p, Dp, b.proj, M, h, Dp); a.resid = h;
            HIPCHK(launch_gemm(a, EPI_RESID, GEMM_VARIANT_AUTO, s));
        }
        VRCHK(prof_end(m, VR_PROF_VIT_PROJ, 2.0 * M * D * D, s));
        HIPCHK(launch_layernorm(h, M, D, Dp, b.n2w.v.as<float>(), b.n2b.v.as<float>(), c.vit_ln_eps, m->w_xn.p, Dp, s));
        VRCHK(prof_begin(m, VR_PROF_VIT_FC1, s));
        { GemmArgs a = gemm_args(m->w_xn.p, Dp, b.fc1, M, m->w_mlp.p, m->Fp); HIPCHK(launch_gemm(a, EPI_GELU, GEMM_VARIANT_AUTO, s)); }
        VRCHK(prof_end(m, VR_PROF_VIT_FC1, 2.0 * M * D * m->F, s));
        VRCHK(prof_begin(m, VR_PROF_VIT_FC2, s));
        {
            GemmArgs a = gemm_args(m->w_mlp.p, m->Fp, b.fc2, M, h, Dp); a.resid = h;
            HIPCHK(launch_gemm(a, EPI_RESID, GEMM_VARIANT_AUTO, s));
        }
        VRCHK(prof_end(m, VR_PROF_VIT_FC2, 2.0 * M * D * m->F, s));
        if (l == 0 && first_group) VRCHK(tap_store(m, "vit_block0", h, N, D, Dp, false, s));
    }
    HIPCHK(launch_layernorm(h, M, D, Dp, m->vit_nw.v.as<float>(), m->vit_nb.v.as<float>(), c.vit_ln_eps, m->w_xn.p, Dp, s));
    if (first_group) VRCHK(tap_store(m, "vit_out", m->w_xn.p, N, D, Dp, true, s));

    // ---- resampler (resampler.py:146-168)
    VRCHK(prof_begin(m, VR_PROF_RESAMPLER, s));
    { GemmArgs a = gemm_args(m->w_xn.p, Dp, m->r_kvproj, M, m->w_kv32.p, E); HIPCHK(launch_gemm(a, EPI_F32, GEMM_VARIANT_AUTO, s)); }
    HIPCHK(launch_layernorm(m->w_kv32.as<float>(), M, E, E, m->r_lnkv_w.v.as<float>(), m->r_lnkv_b.v.as<float>(),
                            c.resampler_ln_eps, m->w_xkv.p, E, s));
    {   // fused k|v in-projection; the k half gets the per-position term pos_k[row % N]
        GemmArgs a = gemm_args(m->w_xkv.p, E, m->r_kv, M, m->w_KV.p, 2 * E);
        a.rowbias = g->pos_k.as<float>(); a.rowbias_period = N; a.rowbias_ld = E; a.rowbias_cols = E;
        HIPCHK(launch_gemm(a, EPI_BF16, GEMM_VARIANT_AUTO, s));
    }
    {
        AttnArgs a{};
        a.q = m->r_q.p; a.ldq = E; a.k = m->w_KV.p; a.ldk = 2 * E; a.v = (const char*)m->w_KV.p + (size_t)E * 2; a.ldv = 2 * E;
        a.out = m->w_ratt.p; a.ldo = E; a.cu_q = cu_qry; a.cu_kv = cu_tok; a.B = n; a.heads = E / 128; a.head_dim = 128;
        a.max_q = Q; a.causal = 0; a.q_shared = 1; a.scale = 1.0f / sqrtf(128.0f);
        HIPCHK(launch_attention(a, s));
    }
    const int R = n * Q;
    { GemmArgs a = gemm_args(m->w_ratt.p, E, m->r_out, R, m->w_rout.p, E); HIPCHK(launch_gemm(a, EPI_F32, GEMM_VARIANT_AUTO, s)); }
    HIPCHK(launch_layernorm(m->w_rout.as<float>(), R, E, E, m->r_lnpost_w.v.as<float>(), m->r_lnpost_b.v.as<float>(),
                            c.resampler_ln_eps, m->w_rln.p, E, s));
    if (first_group && m->taps_on) {   // un-scattered copy for the tap
        GemmArgs a = gemm_args(m->w_rln.p, E, m->r_proj, Q, m->w_rout.p, E);
        HIPCHK(launch_gemm(a, EPI_F32, GEMM_VARIANT_AUTO, s));
        VRCHK(tap_store(m, "resampler_out", m->w_rout.p, Q, E, E, false, s));
    }
    {   // x @ proj, scattered straight into the decoder's fp32 input rows (image_bound scatter_,
        // modeling_minicpmv.py:148-166; vision rows are NOT scaled by scale_emb)
        GemmArgs a = gemm_args(m->w_rln.p, E, m->r_proj, R, m->w_h.p, E);
        a.rowmap = m->w_rowmap.as<int>();
        HIPCHK(launch_gemm(a, EPI_F32, GEMM_VARIANT_AUTO, s));
    }
    VRCHK(prof_end(m, VR_PROF_RESAMPLER, 2.0 * M * D * E + 4.0 * M * E * (double)E + 4.0 * R * (double)N * E + 4.0 * R * E * (double)E, s));
    return VR_OK;
}

// ---- the decoder pass of a token-only batch at fp32-class precision (hp_text.hip) -----------------------------
// out (+)= alpha * A W^T with A = hi + lo (bf16 halves a_hi / a_lo, row pitch lda: w_hp_hi holds [hi rows | lo rows]) and W = L (+ Llo):
// A_hi W_hi, then A_lo W_hi and A_hi W_lo added in place by the residual epilogue of the same MFMA kernels.
static int hp_gemm_tiles(vr_model_s* m, const void* a_hi, const void* a_lo, const Linear& L, const Linear& Llo, int lda, int T, float* out,
                         int ldo, bool into_resid, float alpha, hipStream_t s) {
    // Round 4: the hi and lo halves of A are the ROWS of one GEMM — the lo rows are laid out right behind the T hi rows
    // (run_decoder_hp) — so W_hi is streamed once, M = 2 T rows fill the 256-row tiles (T = 1300 for 64 queries ran 128 x 128
    // tiles at 257 TF as two launches with a read-modify-write epilogue), and one fixed-order sum closes the planes
    // [A_hi W_hi | A_lo W_hi | A_hi W_lo].  Falls back to the three in-place launches when the planes do not fit.
    const size_t plane = (size_t)T * L.n_pad;
    const int n_planes = Llo.has_w ? 3 : 2;
    // (a Linear with a bias — none in the MiniCPM decoder — keeps the three in-place launches, whose first one adds it once: the plane
    // sum has no bias term.  ADVICE r4)
    if (!L.has_b && (const char*)a_lo == (const char*)a_hi + (size_t)T * lda * 2 && m->w_hp_planes.reserve((size_t)n_planes * plane * 4) == VR_OK) {
        float* part = m->w_hp_planes.as<float>();
        {
            GemmArgs a = gemm_args(a_hi, lda, L, 2 * T, part, L.n_pad);
            a.bias = nullptr;
            HIPCHK(launch_gemm(a, EPI_F32, GEMM_VARIANT_AUTO, s));
        }
        if (Llo.has_w) {
            GemmArgs a = gemm_args(a_hi, lda, Llo, T, part + 2 * plane, L.n_pad);
            a.bias = nullptr;
            HIPCHK(launch_gemm(a, EPI_F32, GEMM_VARIANT_AUTO, s));
        }
        HIPCHK(launch_planes_sum(part, n_planes, plane, L.n_pad, T, std::min(L.n_pad, ldo), out, ldo, into_resid ? alpha : 1.0f, into_resid, s));
        return VR_OK;
    }
    {
        GemmArgs a = gemm_args(a_hi, lda, L, T, out, ldo);
        if (into_resid) { a.resid = out; a.alpha = alpha; }
        HIPCHK(launch_gemm(a, into_resid ? EPI_RESID : EPI_F32, GEMM_VARIANT_AUTO, s));
    }
    {
        GemmArgs a = gemm_args(a_lo, lda, L, T, out, ldo);
        a.bias = nullptr;                                   // (added by the launch above)
        a.resid = out; a.alpha = into_resid ? alpha : 1.0f;
        HIPCHK(launch_gemm(a, EPI_RESID, GEMM_VARIANT_AUTO, s));
    }
    if (Llo.has_w) {
        GemmArgs a = gemm_args(a_hi, lda, Llo, T, out, ldo);
        a.bias = nullptr;
        a.resid = out; a.alpha = into_resid ? alpha : 1.0f;
        HIPCHK(launch_gemm(a, EPI_RESID, GEMM_VARIANT_AUTO, s));
    }
    return VR_OK;
}

// The same product for a SHORT batch (T <= 32 rows: one query, a few queries): the tile GEMMs above run a 128-row tile with
// a row in five real and stream a projection's weights through ~50 workgroups (0.55 TB/s: 20 ms for one query, 11 GB of
// weights).  Here every pass is ONE launch of the weight streamer (gemm_skinny.hip, 4-6 TB/s) writing fp32 split-K planes —
// A_hi W_hi | A_lo W_hi | A_hi W_lo into consecutive plane groups — and one launch sums them (fixed order) into `out`.
static int choose_stream_ksplit(int n_pad, int k_pad) {
    const int tiles = (n_pad + 255) / 256, nk = k_pad / 64;
    int best = 1;
    for (int d = 2; d <= 32 && tiles * d <= 256; ++d) {
        const int per = (nk + d - 1) / d;
        if (per >= 4 && (d - 1) * per < nk) best = d;
    }
    return best;
}
static int hp_gemm_stream(vr_model_s* m, const void* a_hi, const void* a_lo, const Linear& L, const Linear& Llo, int lda, int T, float* out,
                          int ldo, bool into_resid, float alpha, hipStream_t s) {
    const int ks = choose_stream_ksplit(L.n_pad, L.k_pad);
    const size_t plane = (size_t)T * L.n_pad;
    const int n_pass = Llo.has_w ? 3 : 2;
    if ((size_t)n_pass * ks * plane * 4 > m->w_hp_part.bytes) return fail(VR_ERR_CAPACITY, "split-K planes exceed the workspace");
    float* part = m->w_hp_part.as<float>();
    for (int pass = 0; pass < n_pass; ++pass) {
        GemmArgs a = gemm_args(pass == 1 ? a_lo : a_hi, lda, pass == 2 ? Llo : L, T, part + (size_t)pass * ks * plane, L.n_pad);
        a.bias = nullptr;
        a.ksplit = ks; a.split_stride = plane;
        HIPCHK(launch_gemm_skinny(a, s));
    }
    HIPCHK(launch_planes_sum(part, n_pass * ks, plane, L.n_pad, T, std::min(L.n_pad, ldo), out, ldo, into_resid ? alpha : 1.0f, into_resid, s));
    return VR_OK;
}
static bool hp_stream_ok(const vr_model_s* m, int T) {
    if (T > 32) return false;
    for (const DecLayer& L : m->layers)
        for (const Linear* l : {&L.qkv, &L.o, &L.gu, &L.down})
            if (l->n_pad % 256 || l->k_pad % 64 || l->has_b) return false;     // 256-column tiles of the streamer; no bias in this decoder
    return true;
}

static int run_decoder_hp(vr_model_s* m, int T, int B, int max_len, hipStream_t s) {
    (void)max_len;
    const vr_config_t& c = m->c;
    const int E = m->E, I = m->I, Ip = m->Ip;
    float* h = m->w_h.as<float>();
    float* qkv = m->w_hp_qkv.as<float>();
    float* att = m->w_hp_att.as<float>();
    float* gu = m->w_hp_gu.as<float>();
    const int ld_gu = pad128(2 * I);
    HIPCHK(launch_seq_of(m->w_seq.as<int>(), B, m->w_seqof.as<int>(), s));
    const bool stream = hp_stream_ok(m, T);
    void* const hi = m->w_hp_hi.p;
    auto lo = [&](int lda) -> void* { return (char*)hi + (size_t)T * lda * 2; };     // the lo rows: right behind the T hi rows
    auto hp_gemm = [&](vr_model_s* mm, const Linear& L, const Linear& Llo, int lda, int TT, float* out, int ldo, bool into_resid,
                       float alpha, hipStream_t ss) -> int {
        return stream ? hp_gemm_stream(mm, hi, lo(lda), L, Llo, lda, TT, out, ldo, into_resid, alpha, ss)
                      : hp_gemm_tiles(mm, hi, lo(lda), L, Llo, lda, TT, out, ldo, into_resid, alpha, ss);
    };
    for (int l = 0; l < c.num_layers; ++l) {
        const DecLayer& L = m->layers[l];
        HIPCHK(launch_rmsnorm_split(h, T, E, L.ln1.v.as<float>(), c.rms_norm_eps, hi, lo(E), s));
        VRCHK(hp_gemm(m, L.qkv, L.qkv_lo, E, T, qkv, 3 * E, false, 1.0f, s));
        HIPCHK(launch_rope_f32(qkv, T, 3 * E, 2 * E, m->w_pos.as<int>(), m->rope.as<float>(), s));
        HIPCHK(launch_attn_f32(qkv, 3 * E, E, m->w_seqof.as<int>(), m->w_seq.as<int>(), T, c.num_heads, 1.0f / sqrtf(64.0f), att, s));
        HIPCHK(launch_split_bf16(att, hi, lo(E), (size_t)T * E, s));
        VRCHK(hp_gemm(m, L.o, L.o_lo, E, T, h, E, true, c.residual_scale, s));
        HIPCHK(launch_rmsnorm_split(h, T, E, L.ln2.v.as<float>(), c.rms_norm_eps, hi, lo(E), s));
        VRCHK(hp_gemm(m, L.gu, L.gu_lo, E, T, gu, ld_gu, false, 1.0f, s));
        HIPCHK(launch_swiglu_split(gu, T, ld_gu, I, Ip, hi, lo(Ip), s));
        VRCHK(hp_gemm(m, L.down, L.down_lo, Ip, T, h, E, true, c.residual_scale, s));
        if (l == 0) VRCHK(tap_store(m, "dec_layer0", h, T, E, E, false, s));
    }
    return VR_OK;
}

static int encode_impl(vr_model_t m, const uint8_t* const* slices, const int32_t* slice_hw, int32_t n_slices,
                       int32_t slices_on_device, const int32_t* input_ids, const int32_t* seq_offsets, int32_t B,
                       const int32_t* vision_rows, float* out_reps, int32_t out_on_device, void* stream,
                       float* out_hidden = nullptr, int32_t hidden_len = 0);

extern "C" int vr_encode(vr_model_t m, const uint8_t* const* slices, const int32_t* slice_hw, int32_t n_slices,
                         int32_t slices_on_device, const int32_t* input_ids, const int32_t* seq_offsets, int32_t B,
                         const int32_t* vision_rows, float* out_reps, int32_t out_on_device, void* stream) {
    if (!m) return fail(VR_ERR_INVALID, "NULL model");
    m->arena_open = false;
    const int rc = encode_impl(m, slices, slice_hw, n_slices, slices_on_device, input_ids, seq_offsets, B, vision_rows,
                               out_reps, out_on_device, stream);
    // the pinned arena feeds async H2D copies: on EVERY exit after the first of them (a failure in
    // the middle of the call included) mark when they have run, or the next call would overwrite
    // the arena under copies that are still pending
    if (m->arena_open && m->arena_ev) {
        if (hipEventRecord(m->arena_ev, (hipStream_t)stream) == hipSuccess) m->arena_pending = true;
        else (void)hipStreamSynchronize((hipStream_t)stream);
    }
    m->arena_open = false;
    return rc;
}

// The HF-style forward of the second caller (visrag_scripts/demo/visrag_pipeline/utils.py:12-32 pools `last_hidden_state` itself):
// the same pass, and the post-norm hidden states as a right-padded [B][hidden_len][hidden_size] fp32 device tensor.
extern "C" int vr_encode_hidden(vr_model_t m, const uint8_t* const* slices, const int32_t* slice_hw, int32_t n_slices,
                                int32_t slices_on_device, const int32_t* input_ids, const int32_t* seq_offsets, int32_t B,
                                const int32_t* vision_rows, float* out_reps, int32_t out_on_device, float* out_hidden,
                                int32_t hidden_len, void* stream) {
    if (!m) return fail(VR_ERR_INVALID, "NULL model");
    if (!out_hidden || hidden_len <= 0) return fail(VR_ERR_INVALID, "out_hidden is NULL or hidden_len <= 0");
    m->arena_open = false;
    const int rc = encode_impl(m, slices, slice_hw, n_slices, slices_on_device, input_ids, seq_offsets, B, vision_rows,
                               out_reps, out_on_device, stream, out_hidden, hidden_len);
    if (m->arena_open && m->arena_ev) {
        if (hipEventRecord(m->arena_ev, (hipStream_t)stream) == hipSuccess) m->arena_pending = true;
        else (void)hipStreamSynchronize((hipStream_t)stream);
    }
    m->arena_open = false;
    return rc;
}

static int encode_impl(vr_model_t m, const uint8_t* const* slices, const int32_t* slice_hw, int32_t n_slices,
                       int32_t slices_on_device, const int32_t* input_ids, const int32_t* seq_offsets, int32_t B,
                       const int32_t* vision_rows, float* out_reps, int32_t out_on_device, void* stream,
                       float* out_hidden, int32_t hidden_len) {
    if (!m->finalized) return fail(VR_ERR_STATE, "vr_model_finalize() has not succeeded");
    if (B <= 0 || !input_ids || !seq_offsets || !out_reps) return fail(VR_ERR_INVALID, "empty batch or NULL argument");
    if (n_slices > 0 && (!slices || !slice_hw || !vision_rows)) return fail(VR_ERR_INVALID, "NULL slice arguments");
    VRCHK(set_dev(m->device));
    hipStream_t s = (hipStream_t)stream;
    const vr_config_t& c = m->c;
    const int E = m->E, Q = m->Q;
    const int T = seq_offsets[B];
    if (seq_offsets[0] != 0) return fail(VR_ERR_INVALID, "seq_offsets[0] must be 0");
    if (B > c.max_seqs) return fail(VR_ERR_CAPACITY, "B=%d exceeds max_seqs=%d", B, c.max_seqs);
    if (T > c.max_tokens) return fail(VR_ERR_CAPACITY, "T=%d tokens exceed max_tokens=%d", T, c.max_tokens);
    for (int i = 0; i < B; ++i)
        if (seq_offsets[i + 1] <= seq_offsets[i]) return fail(VR_ERR_INVALID, "empty sequence %d", i);
    for (int t = 0; t < T; ++t)
        if (input_ids[t] < 0 || input_ids[t] >= c.vocab_size) return fail(VR_ERR_INVALID, "token id %d out of range at %d", input_ids[t], t);
    for (int i = 0; i < n_slices * Q; ++i)
        if (vision_rows[i] >= T) return fail(VR_ERR_INVALID, "vision row %d out of range", vision_rows[i]);

    // ---- K13: token embeddings * scale_emb into the fp32 stream
    VRCHK(arena_begin(m, (size_t)T * 4 + (size_t)(B + 1) * 4 + (size_t)n_slices * (Q * 4 + 8 + 64 * 3) + 4096));
    {
        int* a_ids = (int*)arena_take(m, (size_t)T * 4);
        int* a_seq = (int*)arena_take(m, (size_t)(B + 1) * 4);
        memcpy(a_ids, input_ids, (size_t)T * 4);
        memcpy(a_seq, seq_offsets, (size_t)(B + 1) * 4);
        HIPCHK(hipMemcpyAsync(m->w_ids.p, a_ids, (size_t)T * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(m->w_seq.p, a_seq, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, s));
    }
    // token-only batches (queries, text passages) take the split-precision decoder pass (hp_text.hip).
    // text_split_precision: 0 = never, 1 = every token-only batch, N > 1 = only batches whose longest sequence has at most N
    // tokens (e.g. 512 = the reference's query length: long text passages then stay on the bf16 MFMA flash-attention path,
    // whose cost grows with L instead of L^2 waves).  NOTE the batch-composition dependence this implies: the same text item
    // is embedded at fp32-class precision in a token-only batch and at bf16 precision (inside the 1e-3 tolerance, not the
    // 1e-6 of the split pass) in a batch that also holds an image.
    int max_len = 0;
    for (int i = 0; i < B; ++i) max_len = std::max(max_len, seq_offsets[i + 1] - seq_offsets[i]);
    const bool hp = n_slices == 0 && c.text_split_precision != 0 && (c.text_split_precision == 1 || max_len <= c.text_split_precision);
    if (hp && m->has_embed_lo)
        HIPCHK(launch_embed_gather_hp(m->w_ids.as<int>(), T, m->embed.p, m->embed_lo.p, E, c.scale_emb, m->w_h.as<float>(), s));
    else
        HIPCHK(launch_embed_gather(m->w_ids.as<int>(), T, m->embed.p, E, c.scale_emb, m->w_h.as<float>(), s));
    HIPCHK(launch_iota_pos(m->w_seq.as<int>(), B, m->w_pos.as<int>(), s));

    // ---- vision: group slices by shape, chunk to the workspace
    if (n_slices > 0) {
        size_t pix_total = 0;
        std::vector<size_t> pix_off(n_slices);
        for (int i = 0; i < n_slices; ++i) {
            const int H = slice_hw[2 * i], W = slice_hw[2 * i + 1];
            if (H <= 0 || W <= 0 || H % c.patch_size || W % c.patch_size)
                return fail(VR_ERR_INVALID, "slice %d: %dx%d is not a multiple of the patch size", i, H, W);
            if ((H / c.patch_size) * (W / c.patch_size) > c.max_patches)
                return fail(VR_ERR_CAPACITY, "slice %d has more than max_patches=%d patches", i, c.max_patches);
            pix_off[i] = pix_total;
            pix_total += ((size_t)H * W * 3 + 15) / 16 * 16;
        }
        std::vector<const uint8_t*> dev_ptr(n_slices);
        if (slices_on_device) {
            for (int i = 0; i < n_slices; ++i) dev_ptr[i] = slices[i];
        } else {
            if (m->w_pix.bytes < pix_total) VRCHK(m->w_pix.alloc(pix_total));
            for (int i = 0; i < n_slices; ++i) {
                const size_t nb = (size_t)slice_hw[2 * i] * slice_hw[2 * i + 1] * 3;
                HIPCHK(hipMemcpyAsync((char*)m->w_pix.p + pix_off[i], slices[i], nb, hipMemcpyHostToDevice, s));
                dev_ptr[i] = (const uint8_t*)m->w_pix.p + pix_off[i];
            }
        }
        std::vector<char> done(n_slices, 0);
        bool first = true;
        for (int i = 0; i < n_slices; ++i) {
            if (done[i]) continue;
            const int H = slice_hw[2 * i], W = slice_hw[2 * i + 1];
            const int N = (H / c.patch_size) * (W / c.patch_size);
            std::vector<int> idx;
            for (int j = i; j < n_slices; ++j)
                if (!done[j] && slice_hw[2 * j] == H && slice_hw[2 * j + 1] == W) { idx.push_back(j); done[j] = 1; }
            const int per_pass = std::max(1, (int)std::min<int64_t>(c.max_images, ((int64_t)c.max_images * c.max_patches) / N));
            for (size_t lo = 0; lo < idx.size(); lo += per_pass) {
                const int n = (int)std::min<size_t>(per_pass, idx.size() - lo);
                const uint8_t** ptrs = (const uint8_t**)arena_take(m, (size_t)n * 8);
                int32_t* rowmap = (int32_t*)arena_take(m, (size_t)n * Q * 4);
                for (int k = 0; k < n; ++k) {
                    ptrs[k] = dev_ptr[idx[lo + k]];
                    memcpy(rowmap + (size_t)k * Q, vision_rows + (size_t)idx[lo + k] * Q, (size_t)Q * 4);
                }
                VRCHK(run_vision_group(m, ptrs, n, H, W, rowmap, first, s));
                first = false;
            }
        }
    }
    VRCHK(tap_store(m, "inputs_embeds", m->w_h.p, T, E, E, false, s));

    // ---- decoder (modeling_minicpm.py:939-1004, 1147-1304), packed ragged sequences
    float* h = m->w_h.as<float>();
    const int* seq = m->w_seq.as<int>();
    if (max_len > m->rope_len) return fail(VR_ERR_CAPACITY, "sequence of %d tokens exceeds the RoPE table", max_len);
    VRCHK(prof_begin(m, VR_PROF_DECODER, s));
    if (hp) {
        VRCHK(run_decoder_hp(m, T, B, max_len, s));
    } else {
    // o / down projections (N = E): with few rows the 256^2 grid is far smaller than the chip
    // (T = 2176: 81 tiles on 256 CUs; the 128^2 grid of 306 tiles runs 1.2 waves at 550-700 TF).
    // Split K over `ks` workgroups per tile instead (243 workgroups); the fp32 partial products are
    // summed, in a fixed order, by the RMSNorm that follows anyway (launch_rmsnorm_accum), which also
    // applies the scaled residual update — no reduction pass, no atomics, deterministic.
    int ks = 1;
    int proj_variant = GEMM_VARIANT_256W;
    {
#ifndef VR_DEC_PROJ_192
#define VR_DEC_PROJ_192 1
#endif
        // 256 x 192 tiles when E allows (2304 = 12 x 192): 108 tiles x 2 splits = 216 workgroups write TWO partial
        // planes instead of 81 x 3 = 243 writing three — the fp32 partials (and their re-read by the RMSNorm that
        // sums them) are what these launches spend their epilogue on, at the HBM's pace
        const long t192 = (long)((T + 255) / 256) * (E / 192);
        if (VR_DEC_PROJ_192 && E % 192 == 0 && t192 * 2 <= 256 && t192 * 2 > 160 && E % 128 == 0 && m->Ip % 128 == 0) {
            ks = 2;
            proj_variant = GEMM_VARIANT_192W;
        } else {
            const long tiles = (long)((T + 255) / 256) * ((E + 255) / 256);
            for (int cand = DEC_KSPLIT_MAX; cand > 1; --cand)
                if (tiles * cand <= 256 && E % (cand * 64) == 0 && m->Ip % (cand * 64) == 0) { ks = cand; break; }
        }
        if (m->taps_on) ks = 1;          // (taps read h between the projection and the next norm)
    }
    // Round 6: when half-height tiles fill one round of the chip (T = 2176: 17 x 12 = 204 workgroups of 128 x 192) the
    // projections run with the FULL K per workgroup on the three-stage one-wave kernel (gemm128w.hip) and add into the
    // residual stream in place: no fp32 planes, the RMSNorm behind them reads 20 MB instead of moving 88
#ifndef VR_DEC_PROJ_128W
#define VR_DEC_PROJ_128W 1
#endif
    bool proj_128w = false, gu_128w = false;
#ifndef VR_DEC_GU_128W
#define VR_DEC_GU_128W 0
#endif
    {
        // (A/B knob: gate / up on 128 x 256 tiles when the 256-row grid leaves a ragged second round — T = 2176: 405 tiles on
        // 256 CUs against 765 half-height ones)
        const long Ngu = 2L * m->Ip, t256 = (long)((T + 255) / 256) * ((Ngu + 255) / 256), t128g = (long)((T + 127) / 128) * (Ngu / 256);
        gu_128w = VR_DEC_GU_128W && Ngu % 256 == 0 && (t256 % 256) != 0 && (double)t256 / (double)((t256 + 255) / 256 * 256) < 0.85 &&
                  (double)t128g / (double)((t128g + 255) / 256 * 256) > 0.95;
        const long t128 = (long)((T + 127) / 128) * (E / 192);
        proj_128w = VR_DEC_PROJ_128W && E % 192 == 0 && t128 > 128 && t128 <= 256;
        if (proj_128w) ks = 1;
    }
    const size_t pstride = (size_t)T * E;
    float* part = m->w_part.as<float>();
    bool pend = false;                   // h still lacks residual_scale * sum(partials)
#ifndef VR_DEC_O_GLDS
#define VR_DEC_O_GLDS 0
#endif
#ifndef VR_DEC_O_NOSPLIT
#define VR_DEC_O_NOSPLIT 0
#endif
    auto proj = [&](const void* A, int lda, const Linear& L, bool is_o = false) -> int {
        if (proj_128w) {
            GemmArgs a = gemm_args(A, lda, L, T, h, E); a.resid = h; a.alpha = c.residual_scale;
            if (gemm128w_fits(a, 192)) { HIPCHK(launch_gemm(a, EPI_RESID, GEMM_VARIANT_128W_192, s)); return VR_OK; }
        }
        if (VR_DEC_O_NOSPLIT && is_o && ks > 1 && proj_variant == GEMM_VARIANT_192W) {
            // A/B knob: the o projection (K = E: 36 K-steps) WITHOUT split-K on the 256 x 192 tile — 108 workgroups on 256 CUs,
            // but no fp32 planes: the residual epilogue adds in place and the RMSNorm that follows is the plain one
            GemmArgs a = gemm_args(A, lda, L, T, h, E); a.resid = h; a.alpha = c.residual_scale;
            HIPCHK(launch_gemm(a, EPI_RESID, GEMM_VARIANT_192W, s));
            return VR_OK;
        }
        if (VR_DEC_O_GLDS && is_o && !m->taps_on) {
            // A/B knob: the o projection (K = E: the smaller of the two) on 128 x 128 tiles straight into the residual stream
            GemmArgs a = gemm_args(A, lda, L, T, h, E); a.resid = h; a.alpha = c.residual_scale;
            HIPCHK(launch_gemm(a, EPI_RESID, GEMM_VARIANT_GLDS, s));
            return VR_OK;
        }
        if (ks > 1) {
            GemmArgs a = gemm_args(A, lda, L, T, part, E);
            a.ksplit = ks; a.split_stride = pstride;
            HIPCHK(launch_gemm(a, EPI_F32, gemm256w_fits(a, proj_variant == GEMM_VARIANT_192W ? 192 : 256) ? proj_variant : GEMM_VARIANT_256IL, s));
            pend = true;
        } else {
            GemmArgs a = gemm_args(A, lda, L, T, h, E); a.resid = h; a.alpha = c.residual_scale;
            HIPCHK(launch_gemm(a, EPI_RESID, GEMM_VARIANT_AUTO, s));
        }
        return VR_OK;
    };
    auto norm = [&](const float* w) -> int {
        VRCHK(prof_begin(m, VR_PROF_DEC_NORM, s));
        struct End { vr_model_s* m; hipStream_t s; ~End() { (void)prof_end(m, VR_PROF_DEC_NORM, 0.0, s); } } end_{m, s};
        if (pend) {
            HIPCHK(launch_rmsnorm_accum(h, T, E, E, part, ks, pstride, E, c.residual_scale, w, c.rms_norm_eps, m->w_dxn.p, E, s));
            pend = false;
        } else {
            HIPCHK(launch_rmsnorm(h, T, E, E, w, c.rms_norm_eps, m->w_dxn.p, E, s));
        }
        return VR_OK;
    };
    double attn_flops = 0;
    for (int i = 0; i < B; ++i) { const double Li = seq_offsets[i + 1] - seq_offsets[i]; attn_flops += 4.0 * Li * Li * E; }
    for (int l = 0; l < c.num_layers; ++l) {
        const DecLayer& L = m->layers[l];
        VRCHK(norm(L.ln1.v.as<float>()));
        {
            GemmArgs a = gemm_args(m->w_dxn.p, E, L.qkv, T, m->w_dqkv.p, 3 * E);
            a.rope_pos = m->w_pos.as<int>(); a.rope_table = m->rope.as<float>(); a.rope_cols = 2 * E;
            VRCHK(prof_begin(m, VR_PROF_DEC_QKV, s));
            HIPCHK(launch_gemm(a, EPI_ROPE, GEMM_VARIANT_AUTO, s));
            VRCHK(prof_end(m, VR_PROF_DEC_QKV, 6.0 * T * (double)E * E, s));
        }
        {
            VRCHK(prof_begin(m, VR_PROF_DEC_ATTN, s));
            AttnArgs a{};
            a.q = m->w_dqkv.p; a.ldq = 3 * E; a.k = (const char*)m->w_dqkv.p + (size_t)E * 2; a.ldk = 3 * E;
            a.v = (const char*)m->w_dqkv.p + (size_t)2 * E * 2; a.ldv = 3 * E;
            a.out = m->w_datt.p; a.ldo = E; a.cu_q = seq; a.cu_kv = seq; a.B = B; a.heads = c.num_heads; a.head_dim = 64;
            a.max_q = max_len; a.causal = 1; a.q_shared = 0; a.scale = 1.0f / sqrtf(64.0f);
            HIPCHK(launch_attention(a, s));
            VRCHK(prof_end(m, VR_PROF_DEC_ATTN, attn_flops, s));
        }
        VRCHK(prof_begin(m, VR_PROF_DEC_O, s));
        VRCHK(proj(m->w_datt.p, E, L.o, true));
        VRCHK(prof_end(m, VR_PROF_DEC_O, 2.0 * T * (double)E * E, s));
        VRCHK(norm(L.ln2.v.as<float>()));
        VRCHK(prof_begin(m, VR_PROF_DEC_GU, s));
        {
            GemmArgs a = gemm_args(m->w_dxn.p, E, L.gu, T, m->w_dact.p, m->Ip);
            HIPCHK(launch_gemm(a, EPI_SWIGLU, gu_128w && gemm128w_fits(a, 256) ? GEMM_VARIANT_128W_256 : GEMM_VARIANT_AUTO, s));
        }
        VRCHK(prof_end(m, VR_PROF_DEC_GU, 4.0 * T * (double)E * m->I, s));
        VRCHK(prof_begin(m, VR_PROF_DEC_DOWN, s));
        VRCHK(proj(m->w_dact.p, m->Ip, L.down));
        VRCHK(prof_end(m, VR_PROF_DEC_DOWN, 2.0 * T * (double)E * m->I, s));
        if (l == 0) VRCHK(tap_store(m, "dec_layer0", h, T, E, E, false, s));
    }
    if (pend) {                          // last down projection: residual update only
        HIPCHK(launch_rmsnorm_accum(h, T, E, E, part, ks, pstride, E, c.residual_scale, nullptr, c.rms_norm_eps, nullptr, 0, s));
        pend = false;
    }
    }
    {
        double fl = 0;
        for (int i = 0; i < B; ++i) { const double Li = seq_offsets[i + 1] - seq_offsets[i]; fl += 4.0 * Li * Li * E; }
        fl = c.num_layers * (fl + (double)T * (8.0 * E * E + 6.0 * (double)E * m->I));
        VRCHK(prof_end(m, VR_PROF_DECODER, fl, s));
    }
    // ---- K19+K20: final norm + wmean pool + L2 normalise
    float* tap_hidden = nullptr;   // scratch f32 [T][E] for the post-norm hidden states
    if (m->taps_on && (size_t)T * E * 4 <= m->w_kv32.bytes) tap_hidden = m->w_kv32.as<float>();
    if (out_hidden) {              // vr_encode_hidden: the packed rows first (the resampler's scratch when it is large enough)
        if (max_len > hidden_len) return fail(VR_ERR_INVALID, "hidden_len=%d is shorter than the longest sequence (%d)", hidden_len, max_len);
        if (!tap_hidden) {
            if ((size_t)T * E * 4 <= m->w_kv32.bytes) tap_hidden = m->w_kv32.as<float>();
            else { VRCHK(m->w_hidden.reserve((size_t)T * E * 4)); tap_hidden = m->w_hidden.as<float>(); }
        }
    }
    float* dst = out_on_device ? out_reps : m->w_out.as<float>();
    HIPCHK(launch_pool(h, seq, B, E, m->final_norm.v.as<float>(), c.rms_norm_eps, dst, tap_hidden, s, m->pool_mode));
    if (tap_hidden && m->taps_on) VRCHK(tap_store(m, "last_hidden", tap_hidden, T, E, E, false, s));
    if (out_hidden) {              // right padding (modeling_minicpmv.py:440-479 `pad`): item i -> rows [i * hidden_len, + len_i), zeros behind
        HIPCHK(hipMemsetAsync(out_hidden, 0, (size_t)B * hidden_len * E * 4, s));
        for (int i = 0; i < B; ++i)
            HIPCHK(hipMemcpyAsync(out_hidden + (size_t)i * hidden_len * E, tap_hidden + (size_t)seq_offsets[i] * E,
                                  (size_t)(seq_offsets[i + 1] - seq_offsets[i]) * E * 4, hipMemcpyDeviceToDevice, s));
    }
    if (!out_on_device) HIPCHK(hipMemcpyAsync(out_reps, dst, (size_t)B * E * 4, hipMemcpyDeviceToHost, s));
    if (!out_on_device || (n_slices > 0 && !slices_on_device)) HIPCHK(hipStreamSynchronize(s));   // host buffers consumed
    return VR_OK;
}

// A second handle on the SAME weights with its own workspace, per-grid tables, pinned arena and
// profiling state: two batches can then be in flight on two HIP streams (the tails and the
// LayerNorm/epilogue phases of one batch's kernels overlap the other's GEMMs).  The source handle
// must outlive its clones.
extern "C" int vr_model_clone(vr_model_t src, vr_model_t* out) {
    if (!src || !out) return fail(VR_ERR_INVALID, "src/out is NULL");
    if (!src->finalized) return fail(VR_ERR_STATE, "vr_model_clone before vr_model_finalize");
    VRCHK(set_dev(src->device));
    vr_model_s* m = new vr_model_s(*src);          // shallow: the weight buffers are aliased, never freed by the clone
    m->borrowed = true;
    for (DevBuf* b : {&m->w_hvit, &m->w_xn, &m->w_qkv, &m->w_att, &m->w_mlp, &m->w_kv32, &m->w_xkv, &m->w_KV,
                      &m->w_ratt, &m->w_rout, &m->w_rln, &m->w_h, &m->w_dxn, &m->w_part, &m->w_dqkv, &m->w_datt, &m->w_dact,
                      &m->w_cu, &m->w_ids, &m->w_seq, &m->w_pos, &m->w_rowmap, &m->w_imgptr, &m->w_pix, &m->w_out,
                      &m->w_hp_hi, &m->w_hp_planes, &m->w_hp_qkv, &m->w_hp_att, &m->w_hp_gu, &m->w_seqof, &m->w_hp_part, &m->w_hidden}) {
        b->free();                                 // (a non-owning alias after the copy: just forget it)
    }
    m->grids.clear();                              // (entries alias the source's tables; the clone builds its own)
    m->taps.clear(); m->taps_on = false;
    m->prof_on = false;
    for (auto& pc : m->prof) { pc.ev.clear(); pc.used = 0; pc.ms = 0; pc.flops = 0; pc.launches = 0; }
    m->arena = nullptr; m->arena_cap = 0; m->arena_used = 0; m->arena_ev = nullptr; m->arena_pending = false; m->arena_open = false;
    const int r = alloc_workspace(m);
    if (r != VR_OK) { (void)vr_model_destroy(m); return r; }
    *out = m;
    return VR_OK;
}

extern "C" int vr_model_set_profile(vr_model_t m, int32_t enable) {
    if (!m) return fail(VR_ERR_INVALID, "NULL model");
    VRCHK(set_dev(m->device));
    VRCHK(prof_collect(m));
    m->prof_on = enable != 0;
    m->prof_level = enable == 2 ? 2 : (enable ? 1 : 0);
    for (auto& p : m->prof) { p.ms = 0; p.flops = 0; p.launches = 0; p.used = 0; }
    return VR_OK;
}

extern "C" int vr_model_get_profile(vr_model_t m, int32_t cls, double* total_ms, int64_t* launches, double* total_flops) {
    if (!m || cls < 0 || cls >= VR_PROF_CLASSES || !total_ms || !launches || !total_flops)
        return fail(VR_ERR_INVALID, "bad profile arguments");
    VRCHK(set_dev(m->device));
    VRCHK(prof_collect(m));
    *total_ms = m->prof[cls].ms; *launches = m->prof[cls].launches; *total_flops = m->prof[cls].flops;
    return VR_OK;
}

// --------------------------------------------------------------------------------- index ---
constexpr int CERT_WORDS = 32, CERT_FLAG = 2, CERT_FLAG2 = 3, CERT_STATS = 4, CERT_NSTATS = 6;   // (the two flag counters are consecutive: cleared together)
struct vr_index_s {
    int device = 0, dim = 0;
    int64_t cap = 0, n = 0;
    DevBuf f32, bf16;                 // [cap_pad][dim]
    DevBuf q32, qbf, cs, ci, ck, os, oi, ok, thr, sbuf;  // query staging / candidates / outputs / thresholds / score rows
    int64_t qcap = 0, ccap = 0, kcap = 0;
    // certification state (search_common.h), 32 words: f32 [0] largest row norm, [1] largest bf16 rounding residual of a row;
    // int [2] flag count, [3] second-level flag count (exact fp32 pass); u32 [4..9] query counters {certified at once, after
    // extended re-scoring, flagged, uncertified mode, candidates gathered a second time, of the flagged: exact fp32 pass}
    DevBuf cert, flags, flagq;        // flags: int flag_list[fcap] | int flag2_list[fcap] | f32 flag_tau[fcap]; flagq: bf16 [fcap][dim]
    int64_t fcap = 0;
    int* huge_seen = nullptr;         // pinned host word: a streaming search met a band beyond search_band_max() rows (SearchArgs::huge_seen)
    float eps_rel = -2.f;             // -2: the rigorous data-dependent default; >= 0: the caller's eps_rel |q| max|d|; else off
    // per-stage HIP events (vr_index_set_search_profile): convert | thresholds | sweep | merge | exact pass
    bool prof_on = false;
    hipEvent_t prof_ev[SEARCH_PROF_EVENTS] = {};
    double prof_ms[SEARCH_PROF_EVENTS - 1] = {};
    int64_t prof_calls = 0;
};

extern "C" int vr_index_create(int device_id, int32_t dim, int64_t capacity, vr_index_t* out) {
    if (!out || dim <= 0 || capacity <= 0) return fail(VR_ERR_INVALID, "bad index arguments");
    if (dim % 64 || dim > 2560) return fail(VR_ERR_INVALID, "dim %d must be a multiple of 64 and <= 2560", dim);
    if (capacity >= ((int64_t)1 << 31) - 256) return fail(VR_ERR_INVALID, "capacity too large for 32-bit row ids");
    VRCHK(set_dev(device_id));
    vr_index_s* ix = new vr_index_s();
    ix->device = device_id; ix->dim = dim; ix->cap = capacity;
    const int64_t cp = pad256l(capacity);
    int r = ix->f32.alloc((size_t)cp * dim * 4);
    if (r == VR_OK) r = ix->bf16.alloc((size_t)cp * dim * 2);
    if (r == VR_OK) r = ix->cert.alloc(CERT_WORDS * 4);
    if (r == VR_OK && hipHostMalloc((void**)&ix->huge_seen, 64, hipHostMallocDefault) != hipSuccess) r = fail(VR_ERR_HIP, "hipHostMalloc");
    if (r != VR_OK) { ix->f32.free(); ix->bf16.free(); ix->cert.free(); delete ix; return r; }
    *ix->huge_seen = 0;
    *out = ix;
    return VR_OK;
}

extern "C" int vr_index_destroy(vr_index_t ix) {
    if (!ix) return VR_OK;
    (void)hipSetDevice(ix->device);
    (void)hipDeviceSynchronize();
    for (DevBuf* b : {&ix->f32, &ix->bf16, &ix->q32, &ix->qbf, &ix->cs, &ix->ci, &ix->ck, &ix->os, &ix->oi, &ix->ok, &ix->thr,
                      &ix->sbuf, &ix->cert, &ix->flags, &ix->flagq})
        b->free();
    for (hipEvent_t e : ix->prof_ev) if (e) (void)hipEventDestroy(e);
    if (ix->huge_seen) (void)hipHostFree(ix->huge_seen);
    delete ix;
    return VR_OK;
}

extern "C" int vr_index_reset(vr_index_t ix) {
    if (!ix) return fail(VR_ERR_INVALID, "NULL index");
    VRCHK(set_dev(ix->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemset(ix->cert.p, 0, 8));          // largest row norm, largest rounding residual
    ix->n = 0;
    if (ix->huge_seen) *ix->huge_seen = 0;
    return VR_OK;
}

extern "C" int vr_index_size(vr_index_t ix, int64_t* n) {
    if (!ix || !n) return fail(VR_ERR_INVALID, "NULL argument");
    *n = ix->n;
    return VR_OK;
}

extern "C" int vr_index_add(vr_index_t ix, const float* reps, int64_t n, int32_t on_device, void* stream) {
    if (!ix || (!reps && n > 0) || n < 0) return fail(VR_ERR_INVALID, "bad arguments");
    if (n == 0) return VR_OK;
    if (ix->n + n > ix->cap) return fail(VR_ERR_CAPACITY, "index capacity %lld exceeded", (long long)ix->cap);
    VRCHK(set_dev(ix->device));
    hipStream_t s = (hipStream_t)stream;
    float* dst = ix->f32.as<float>() + (size_t)ix->n * ix->dim;
    HIPCHK(hipMemcpyAsync(dst, reps, (size_t)n * ix->dim * 4, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    HIPCHK(launch_f32_to_bf16(dst, (char*)ix->bf16.p + (size_t)ix->n * ix->dim * 2, (size_t)n * ix->dim, s));
    HIPCHK(launch_row_norm_max(dst, n, ix->dim, ix->cert.as<float>(), s));      // max |d|, max |d - bf16(d)|: the search's error bound
    if (!on_device) HIPCHK(hipStreamSynchronize(s));
    ix->n += n;
    return VR_OK;
}

extern "C" int vr_index_set_search_eps(vr_index_t ix, float eps_rel) {
    if (!ix) return fail(VR_ERR_INVALID, "NULL index");
    if (eps_rel != eps_rel) ix->eps_rel = -2.f;    // NaN: back to the default bound
    else ix->eps_rel = eps_rel < 0.f ? -1.f : eps_rel;
    return VR_OK;
}

extern "C" int vr_index_search_stats(vr_index_t ix, int64_t* out6, int32_t reset) {     // out6: SIX words, see the header
    if (!ix || !out6) return fail(VR_ERR_INVALID, "NULL argument");
    VRCHK(set_dev(ix->device));
    HIPCHK(hipDeviceSynchronize());
    unsigned w[CERT_NSTATS];
    HIPCHK(hipMemcpy(w, ix->cert.as<unsigned>() + CERT_STATS, CERT_NSTATS * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < CERT_NSTATS; ++i) out6[i] = w[i];
    if (reset) HIPCHK(hipMemset(ix->cert.as<unsigned>() + CERT_STATS, 0, CERT_NSTATS * 4));
    return VR_OK;
}

extern "C" int vr_index_search_plan(vr_index_t ix, int32_t nq, int32_t* out5) {
    int32_t* out4 = out5;
    if (!ix || !out4 || nq <= 0) return fail(VR_ERR_INVALID, "bad arguments");
    const bool stream = search_uses_stream(nq, ix->dim);
    const int own = stream ? 0 : search_prepass_owned(ix->n, nq, ix->dim);
    const int sweep = stream ? search_stream_chunks() : search_num_chunks(ix->n - (own ? (int64_t)SEARCH_PRE_SPOTS * 256 : 0), nq);
    const int64_t tile = search_uses_256(nq) ? 256 : 128;
    const int64_t tiles = (ix->n + tile - 1) / tile - (own ? SEARCH_PRE_SPOTS : 0);
    out4[0] = sweep + own; out4[1] = own; out4[2] = sweep; out4[3] = stream ? 0 : (int)((tiles + sweep - 1) / sweep);
    out5[4] = (ix->huge_seen && __atomic_load_n(ix->huge_seen, __ATOMIC_RELAXED) != 0) ? 1 : 0;
    return VR_OK;
}

extern "C" int vr_index_error_model(vr_index_t ix, float* out4) {
    if (!ix || !out4) return fail(VR_ERR_INVALID, "NULL argument");
    VRCHK(set_dev(ix->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out4, ix->cert.p, 8, hipMemcpyDeviceToHost));
    out4[2] = search_acc_rel(ix->dim);
    out4[3] = search_default_eps_rel(ix->dim);
    return VR_OK;
}

extern "C" int vr_index_set_search_profile(vr_index_t ix, int32_t enable) {
    if (!ix) return fail(VR_ERR_INVALID, "NULL index");
    VRCHK(set_dev(ix->device));
    if (enable)
        for (hipEvent_t& e : ix->prof_ev) if (!e) HIPCHK(hipEventCreate(&e));
    ix->prof_on = enable != 0;
    for (double& m : ix->prof_ms) m = 0;
    ix->prof_calls = 0;
    return VR_OK;
}

extern "C" int vr_index_get_search_profile(vr_index_t ix, double* ms5, int64_t* calls) {
    if (!ix || !ms5 || !calls) return fail(VR_ERR_INVALID, "NULL argument");
    for (int i = 0; i < SEARCH_PROF_EVENTS - 1; ++i) ms5[i] = ix->prof_ms[i];
    *calls = ix->prof_calls;
    return VR_OK;
}

// queries [nq][dim] -> top k per query, as (scores, ids) or as packed keys with `id_offset` added to the row ids
static int search_impl(vr_index_t ix, const float* queries, int32_t nq, int32_t k, float* out_scores, int64_t* out_ids,
                       unsigned long long* out_keys, int64_t id_offset, int32_t on_device, void* stream) {
    const bool keys_out = out_keys != nullptr;
    if (!ix || !queries || nq <= 0 || (!keys_out && (!out_scores || !out_ids))) return fail(VR_ERR_INVALID, "bad arguments");
    const bool bigk = k > 26;             // deep retrieval: GEMM + radix select (search_bigk.hip)
    if (k <= 0 || k > search_bigk_max()) return fail(VR_ERR_INVALID, "k=%d unsupported (1..%d)", k, search_bigk_max());
    if (keys_out && (id_offset < 0 || id_offset + ix->n >= ((int64_t)1 << 32) - 1))
        return fail(VR_ERR_INVALID, "id_offset %lld + %lld rows do not fit 32-bit global ids", (long long)id_offset, (long long)ix->n);
    const int kp = bigk ? 0 : search_kprime(k);
    VRCHK(set_dev(ix->device));
    hipStream_t s = (hipStream_t)stream;
    const int dim = ix->dim;
    const int64_t ldS = pad256l(std::max<int64_t>(ix->n, 1));
    // queries per pass: the deep path's GEMM writes one fp32 score row per query; the fused path's candidate scratch grows
    // with the pass (128 KiB of half-lists per query at 100k rows)
    int64_t qblk = bigk ? 256 : 4096;
    const int64_t nqp = pad256l(std::min<int64_t>(nq, qblk));
    // score rows of the fallback passes (band pass / exact pass over the FLAGGED queries, search_band.hip): a bounded buffer
    // — at most 512 MiB of fp32 score rows (never fewer than 16 rows) — walked in passes of `slots` flagged queries
    const int64_t slots = bigk ? 256 : std::min<int64_t>(nqp, std::max<int64_t>(16, (((int64_t)1 << 27) / ldS) / 16 * 16));
    if (ix->qcap < nqp) {
        VRCHK(ix->qbf.alloc((size_t)nqp * dim * 2));
        VRCHK(ix->thr.alloc((size_t)nqp * 8));            // thresholds | what the lists are complete down to (thr_cert)
        ix->qcap = nqp;
    }
    const float* q32 = queries;
    if (!on_device) {
        VRCHK(ix->q32.reserve((size_t)nq * dim * 4));
        HIPCHK(hipMemcpyAsync(ix->q32.p, queries, (size_t)nq * dim * 4, hipMemcpyHostToDevice, s));
        q32 = ix->q32.as<float>();
    }
    float* os = out_scores; int64_t* oi = out_ids; unsigned long long* ok = out_keys;
    if (!on_device) {
        if (keys_out) { VRCHK(ix->ok.reserve((size_t)nq * k * 8)); ok = ix->ok.as<unsigned long long>(); }
        else {
            VRCHK(ix->os.reserve((size_t)nq * k * 4));
            VRCHK(ix->oi.reserve((size_t)nq * k * 8));
            os = ix->os.as<float>(); oi = ix->oi.as<int64_t>();
        }
    }
    if (ix->n == 0) {
        if (keys_out) HIPCHK(hipMemsetAsync(ok, 0, (size_t)nq * k * 8, s));
        else {
            std::vector<float> sc((size_t)nq * k, -INFINITY);
            std::vector<int64_t> id((size_t)nq * k, -1);
            HIPCHK(hipMemcpyAsync(os, sc.data(), sc.size() * 4, hipMemcpyHostToDevice, s));
            HIPCHK(hipMemcpyAsync(oi, id.data(), id.size() * 8, hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
        }
    } else {
        if (ix->fcap < nqp) {
            VRCHK(ix->flags.alloc((size_t)nqp * 12));
            VRCHK(ix->flagq.alloc((size_t)(nqp + 256) * dim * 2));     // (+ one tile: the GEMM's last row tile may start anywhere)
            ix->fcap = nqp;
        }
        VRCHK(ix->sbuf.reserve((size_t)slots * ldS * 4));
        for (int64_t q0 = 0; q0 < nq; q0 += qblk) {
            const int nb = (int)std::min<int64_t>(qblk, nq - q0);
            const int64_t nbp = pad256l(nb);
            const bool prof = ix->prof_on && !bigk;
            if (prof) HIPCHK(hipEventRecord(ix->prof_ev[0], s));
            // rows >= nb: zeros; also clears the flag counters.  A handful of queries (streaming kernel): converted inside it.
            const bool conv_in_kernel = !bigk && search_uses_stream(nb, dim);
            if (!conv_in_kernel)
                HIPCHK(launch_f32_to_bf16_pad(q32 + (size_t)q0 * dim, ix->qbf.p, (size_t)nb * dim, (size_t)nbp * dim, s,
                                              ix->cert.as<int>() + CERT_FLAG));   // (clears both flag counters)
            if (prof) HIPCHK(hipEventRecord(ix->prof_ev[1], s));
            SearchArgs a{};
            a.index_bf16 = ix->bf16.p; a.index_f32 = ix->f32.as<float>(); a.n_docs = ix->n; a.dim = dim;
            a.q_bf16 = ix->qbf.p; a.q_f32 = q32 + (size_t)q0 * dim; a.nq = nb; a.k = k;
            a.convert_q = conv_in_kernel ? 1 : 0;
            // (a handful of queries: the streaming sweep leaves every bf16 score behind — a query its merge cannot certify is
            // redone by that merge workgroup itself, and none of the fallback launches below is issued)
            if (conv_in_kernel) { a.score_rows = ix->sbuf.as<float>(); a.ld_scores = (size_t)ldS; }
            // (... unless the band is beyond search_band_max() rows: the first such query is walked by its one workgroup and sets
            // the host-visible word; from then on the exact pass is launched behind this index's streaming searches)
            const bool huge_seen = ix->huge_seen && __atomic_load_n(ix->huge_seen, __ATOMIC_RELAXED) != 0;
            const bool exact_small = conv_in_kernel && huge_seen;
            // (the sweeps behind a pre-pass likewise: band_select_kernel walks the first band beyond search_band_max() rows itself;
            // deep retrieval, k > 26, keeps the exact pass — its select handles any k)
            const bool exact_big = !conv_in_kernel && (bigk || huge_seen || !ix->huge_seen);
            a.exact_follows = (exact_small || exact_big) ? 1 : 0;
            a.huge_seen = ix->huge_seen;
            a.eps_data = ix->eps_rel == -2.f ? 1 : 0;
            a.eps_rel = a.eps_data ? 0.f : ix->eps_rel;
            a.acc_rel = search_acc_rel(dim);
            a.dmax = ix->cert.as<float>();
            a.flag_count = ix->cert.as<int>() + CERT_FLAG; a.flag_list = ix->flags.as<int>();
            a.flag2_count = ix->cert.as<int>() + CERT_FLAG2; a.flag2_list = ix->flags.as<int>() + ix->fcap;
            a.flag_tau = ix->flags.as<float>() + 2 * ix->fcap; a.flag_q = ix->flagq.p;
            a.stats = ix->cert.as<unsigned>() + CERT_STATS;
            if (keys_out) { a.out_keys = ok + (size_t)q0 * k; a.id_offset = id_offset; }
            else { a.out_scores = os + (size_t)q0 * k; a.out_ids = oi + (size_t)q0 * k; }
            a.prof_ev = prof ? ix->prof_ev : nullptr;
            if (bigk) {
                // score rows of <= 256 queries at a time: S[q][doc] = queries x index^T on the bf16 MFMA GEMM
                GemmArgs g{};
                g.A = ix->qbf.p; g.lda = dim;
                g.W = ix->bf16.p; g.ldw = dim; g.M = nb; g.N = (int)pad128l(ix->n); g.K = dim;
                g.out = ix->sbuf.p; g.ldo = (int)ldS; g.alpha = 1.0f;
                HIPCHK(launch_gemm(g, EPI_F32, GEMM_VARIANT_AUTO, s));
                HIPCHK(launch_search_bigk(a, ix->sbuf.as<float>(), (size_t)ldS, 0, nb, s));
            } else {
                // (the 256-tile sweep over >= 128 index tiles: the threshold pre-pass owns its 16 sampled tiles — scored once, its
                // survivors in 8 list chunks of their own behind the sweep's — and the sweep walks the rest: search.hip)
                a.pre_own_chunks = search_uses_stream(nb, dim) ? 0 : search_prepass_owned(ix->n, nb, dim);
                a.n_chunks = search_uses_stream(nb, dim) ? search_stream_chunks()
                             : search_num_chunks(ix->n - (a.pre_own_chunks ? (int64_t)SEARCH_PRE_SPOTS * 256 : 0), nb) + a.pre_own_chunks;
                a.thr_init = ix->thr.as<float>();
                a.thr_cert = ix->thr.as<float>() + ix->qcap;
                const int64_t need = std::max<int64_t>(nbp * a.n_chunks * kp, nbp * search_prepass_floats());
                if (ix->ccap < need) {
                    VRCHK(ix->cs.alloc((size_t)need * 4));
                    VRCHK(ix->ci.alloc((size_t)need * 4));
                    ix->ccap = need;
                }
                a.cand_scores = ix->cs.as<float>(); a.cand_ids = ix->ci.as<int>();
                if (search_uses_256(nb)) {
                    const int64_t kneed = nbp * a.n_chunks * 128;      // [q][chunk][2 halves][64]
                    if (ix->kcap < kneed) { VRCHK(ix->ck.alloc((size_t)kneed * 8)); ix->kcap = kneed; }
                    a.cand_keys = ix->ck.as<unsigned long long>();
                }
                HIPCHK(launch_search(a, s));
            }
            if (prof) HIPCHK(hipEventRecord(ix->prof_ev[4], s));
            if ((a.eps_data || a.eps_rel >= 0.f) && !a.score_rows) {
                // whatever the merge flagged (nothing, normally: every kernel below leaves at once), `slots` queries per pass:
                // bf16 score rows of the flagged queries (GEMM over their compacted bf16 rows, row count on the device) ->
                // every row inside a query's error band re-scored in fp32 (search_band.hip) -> what is left (bands beyond
                // search_band_max() rows) through the exact fp32 pass over the whole index (search_exact.hip)
                for (int64_t f0 = 0; f0 < nb; f0 += slots) {
                    const int ns = (int)std::min<int64_t>(slots, nb - f0);
                    GemmArgs g{};
                    g.A = (const char*)ix->flagq.p + (size_t)f0 * dim * 2; g.lda = dim;
                    g.W = ix->bf16.p; g.ldw = dim; g.M = ns; g.N = (int)pad128l(ix->n); g.K = dim;
                    g.out = ix->sbuf.p; g.ldo = (int)ldS; g.alpha = 1.0f;
                    g.m_dev = a.flag_count; g.m_sub = (int)f0;
                    HIPCHK(launch_gemm(g, EPI_F32, GEMM_VARIANT_256IL, s));
                    HIPCHK(launch_band_select(a, ix->sbuf.as<float>(), (size_t)ldS, (int)f0, ns, s));
                }
                SearchArgs ax = a;
                ax.flag_count = a.flag2_count; ax.flag_list = a.flag2_list;
                for (int64_t f0 = 0; exact_big && f0 < nb; f0 += slots) {
                    const int ns = (int)std::min<int64_t>(slots, nb - f0);
                    HIPCHK(launch_exact_scores(a.index_f32, a.n_docs, dim, a.q_f32, ax.flag_list, ax.flag_count, (int)f0, ns,
                                               ix->sbuf.as<float>(), (size_t)ldS, s));
                    HIPCHK(launch_exact_select(ax, ix->sbuf.as<float>(), (size_t)ldS, (int)f0, ns, s));
                }
            }
            if ((a.eps_data || a.eps_rel >= 0.f) && a.score_rows && exact_small) {
                SearchArgs ax = a;
                ax.flag_count = a.flag2_count; ax.flag_list = a.flag2_list;
                HIPCHK(launch_exact_scores(a.index_f32, a.n_docs, dim, a.q_f32, ax.flag_list, ax.flag_count, 0, nb, ix->sbuf.as<float>(), (size_t)ldS, s));
                HIPCHK(launch_exact_select(ax, ix->sbuf.as<float>(), (size_t)ldS, 0, nb, s));
            }
            if (prof) {
                HIPCHK(hipEventRecord(ix->prof_ev[5], s));
                HIPCHK(hipEventSynchronize(ix->prof_ev[5]));
                for (int i = 0; i + 1 < SEARCH_PROF_EVENTS; ++i) {
                    float ms = 0.f;
                    HIPCHK(hipEventElapsedTime(&ms, ix->prof_ev[i], ix->prof_ev[i + 1]));
                    ix->prof_ms[i] += ms;
                }
                ix->prof_calls += 1;
            }
        }
    }
    if (!on_device) {
        if (keys_out) HIPCHK(hipMemcpyAsync(out_keys, ok, (size_t)nq * k * 8, hipMemcpyDeviceToHost, s));
        else {
            HIPCHK(hipMemcpyAsync(out_scores, os, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(hipMemcpyAsync(out_ids, oi, (size_t)nq * k * 8, hipMemcpyDeviceToHost, s));
        }
        HIPCHK(hipStreamSynchronize(s));
    }
    return VR_OK;
}

extern "C" int vr_index_search(vr_index_t ix, const float* queries, int32_t nq, int32_t k, float* out_scores,
                               int64_t* out_ids, int32_t on_device, void* stream) {
    return search_impl(ix, queries, nq, k, out_scores, out_ids, nullptr, 0, on_device, stream);
}

extern "C" int vr_index_search_keys(vr_index_t ix, const float* queries, int32_t nq, int32_t k, int64_t id_offset,
                                    uint64_t* out_keys, int32_t on_device, void* stream) {
    if (!out_keys) return fail(VR_ERR_INVALID, "out_keys is NULL");
    return search_impl(ix, queries, nq, k, nullptr, nullptr, (unsigned long long*)out_keys, id_offset, on_device, stream);
}

extern "C" int vr_topk_merge(int device_id, const float* scores, const int64_t* ids, int32_t n_parts, int32_t nq,
                             int32_t k, float* out_scores, int64_t* out_ids, void* stream) {
    if (!scores || !ids || !out_scores || !out_ids || n_parts <= 0 || nq <= 0) return fail(VR_ERR_INVALID, "bad arguments");
    VRCHK(set_dev(device_id));
    HIPCHK(launch_topk_merge(scores, ids, n_parts, nq, k, out_scores, out_ids, (hipStream_t)stream));
    return VR_OK;
}

extern "C" int vr_topk_merge_keys(int device_id, const uint64_t* keys, int32_t n_parts, int32_t nq, int32_t k,
                                  float* out_scores, int64_t* out_ids, void* stream) {
    if (!keys || !out_scores || !out_ids || n_parts <= 0 || nq <= 0) return fail(VR_ERR_INVALID, "bad arguments");
    VRCHK(set_dev(device_id));
    HIPCHK(launch_topk_merge_keys((const unsigned long long*)keys, n_parts, nq, k, out_scores, out_ids, (hipStream_t)stream));
    return VR_OK;
}

// -------------------------------------------------------------------------------- resize ---
// coefficient tables are cached per (device, in, out): a corpus has a handful of page sizes
struct ResizeTab { DevBuf bounds, kk; int ksize = 0; };
static std::map<std::tuple<int, int, int>, ResizeTab> g_resize_tabs;

static int get_resize_tab(int dev, int in_size, int out_size, ResizeTab** out) {
    auto key = std::make_tuple(dev, in_size, out_size);
    auto it = g_resize_tabs.find(key);
    if (it == g_resize_tabs.end()) {
        std::vector<int> b, k;
        ResizeTab t;
        t.ksize = resize_coeffs(in_size, out_size, b, k);
        VRCHK(t.bounds.alloc(b.size() * 4));
        VRCHK(t.kk.alloc(k.size() * 4));
        HIPCHK(hipMemcpy(t.bounds.p, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(t.kk.p, k.data(), k.size() * 4, hipMemcpyHostToDevice));
        it = g_resize_tabs.emplace(key, std::move(t)).first;
    }
    *out = &it->second;
    return VR_OK;
}

// grow-only staging buffers per (device, stream): the resize of a page costs no allocation, no
// clear and no stream synchronisation, so it stays asynchronous next to another stream's batch
struct ResizeScratch { DevBuf in, tmp; };
static std::map<std::pair<int, void*>, ResizeScratch> g_resize_scratch;

extern "C" int vr_resize_bicubic(int device_id, const uint8_t* src, int32_t src_on_device, int32_t H, int32_t W,
                                 uint8_t* dst, int32_t out_h, int32_t out_w, void* stream) {
    if (!src || !dst || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) return fail(VR_ERR_INVALID, "bad resize arguments");
    VRCHK(set_dev(device_id));
    hipStream_t s = (hipStream_t)stream;
    ResizeScratch& sc = g_resize_scratch[std::make_pair(device_id, stream)];
    const uint8_t* in = src;
    if (!src_on_device) {
        VRCHK(sc.in.reserve((size_t)H * W * 3));
        HIPCHK(hipMemcpyAsync(sc.in.p, src, (size_t)H * W * 3, hipMemcpyHostToDevice, s));
        in = sc.in.as<uint8_t>();
    }
    const bool need_h = out_w != W, need_v = out_h != H;
    if (!need_h && !need_v) {
        HIPCHK(hipMemcpyAsync(dst, in, (size_t)H * W * 3, hipMemcpyDeviceToDevice, s));
    } else {
        const uint8_t* mid = in;
        if (need_h) {
            ResizeTab* th = nullptr;
            VRCHK(get_resize_tab(device_id, W, out_w, &th));
            uint8_t* hout = dst;
            if (need_v) { VRCHK(sc.tmp.reserve((size_t)H * out_w * 3)); hout = sc.tmp.as<uint8_t>(); }
            HIPCHK(launch_resize_h(in, W, H, hout, out_w, th->bounds.as<int>(), th->kk.as<int>(), th->ksize, s));
            mid = hout;
        }
        if (need_v) {
            ResizeTab* tv = nullptr;
            VRCHK(get_resize_tab(device_id, H, out_h, &tv));
            HIPCHK(launch_resize_v(mid, out_w, dst, out_h, tv->bounds.as<int>(), tv->kk.as<int>(), tv->ksize, s));
        }
    }
    if (!src_on_device) HIPCHK(hipStreamSynchronize(s));     // the caller may reuse its host buffer
    return VR_OK;
}

// ------------------------------------------------------------------------- synthetic input ---
extern "C" int vr_synth_pages(int device_id, uint8_t* out, int32_t n, int32_t size, int64_t seed, int64_t first, void* stream) {
    if (!out || n < 0 || size < 64 || size > 4096) return fail(VR_ERR_INVALID, "bad synth_pages arguments");
    VRCHK(set_dev(device_id));
    HIPCHK(launch_synth_pages(out, n, size, seed, first, (hipStream_t)stream));
    return VR_OK;
}

// ------------------------------------------------------------------- streams and queues ---
// HIP maps a process's streams onto a handful of hardware queues (four by default); two streams that land on ONE queue
// run their kernels one behind the other, and a caller keeping two batches in flight on them gets the single-stream
// rate plus the bookkeeping (measured, round 4: streams 2 and 3 of torch's pool in a fresh process, 661 against 711
// pages/s for every other neighbouring pair).  The mapping is the runtime's business, so it is PROBED: a one-thread
// kernel spins for `usec` on each stream; together they take `usec` if the streams overlap and twice that if not.
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();                     // (the constant 100 MHz counter)
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" int vr_streams_overlap(int device_id, void* stream_a, void* stream_b, int32_t* overlap) {
    if (!overlap) return fail(VR_ERR_INVALID, "NULL argument");
    VRCHK(set_dev(device_id));
    const hipStream_t a = (hipStream_t)stream_a, b = (hipStream_t)stream_b;
    constexpr int usec = 400;
    int votes = 0;
    for (int rep = 0; rep < 3; ++rep) {                      // (first round: also the kernel's load)
        HIPCHK(hipStreamSynchronize(a));
        HIPCHK(hipStreamSynchronize(b));
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, a, (long long)usec * 100);
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, b, (long long)usec * 100);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(a));
        HIPCHK(hipStreamSynchronize(b));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (rep > 0) votes += us < 1.6 * usec ? 1 : 0;
    }
    *overlap = votes == 2 ? 1 : 0;
    return VR_OK;
}

// ------------------------------------------------------------------------------ op-level ---
extern "C" int vr_op_gemm(int device_id, const void* A, int32_t lda, const void* W, int32_t ldw, int32_t M, int32_t N,
                          int32_t K, int32_t epilogue, const float* bias, const float* resid, float alpha, void* out,
                          int32_t ldo, const int32_t* rope_pos, const float* rope_table, int32_t rope_cols,
                          int32_t variant, void* stream) {
    if (!A || !W || !out) return fail(VR_ERR_INVALID, "NULL argument");
    if (variant != GEMM_VARIANT_GLDS && variant != GEMM_VARIANT_AUTO && variant != GEMM_VARIANT_192 && variant != GEMM_VARIANT_256IL && variant != GEMM_VARIANT_256W && variant != GEMM_VARIANT_192W &&
        variant != GEMM_VARIANT_128W_192 && variant != GEMM_VARIANT_128W_256)
        return fail(VR_ERR_INVALID, "variant %d: 0 (128^2 tile), 3 (auto), 7 (256x192 tile), 9 (256^2 tile), 12 / 13 (256^2 / 256x192 tile, one wave per SIMD), 14 / 15 (128x192 / 128x256 tile, one wave per SIMD)", variant);
    if (((variant == 7 || variant == 13 || variant == 14) ? N % 192 : N % 128) || K % 64 || M <= 0) return fail(VR_ERR_INVALID, "need N %% 128 == 0 (192 for variant 7), K %% 64 == 0");
    if (epilogue == EPI_RESID && !resid) return fail(VR_ERR_INVALID, "EPI_RESID needs resid");
    if (epilogue == EPI_ROPE && (!rope_pos || !rope_table)) return fail(VR_ERR_INVALID, "EPI_ROPE needs tables");
    VRCHK(set_dev(device_id));
    GemmArgs a{};
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.M = M; a.N = N; a.K = K; a.bias = bias; a.resid = resid;
    a.alpha = alpha; a.out = out; a.ldo = ldo; a.rope_pos = rope_pos; a.rope_table = rope_table; a.rope_cols = rope_cols;
    HIPCHK(launch_gemm(a, epilogue, variant, (hipStream_t)stream));
    return VR_OK;
}

extern "C" int vr_op_norm(int device_id, int32_t kind, const float* x, int32_t rows, int32_t dim, const float* weight,
                          const float* bias, float eps, void* out, int32_t ldo, void* stream) {
    if (!x || !weight || !out || (kind == 0 && !bias)) return fail(VR_ERR_INVALID, "NULL argument");
    VRCHK(set_dev(device_id));
    if (kind == 0) HIPCHK(launch_layernorm(x, rows, dim, dim, weight, bias, eps, out, ldo, (hipStream_t)stream));
    else HIPCHK(launch_rmsnorm(x, rows, dim, dim, weight, eps, out, ldo, (hipStream_t)stream));
    return VR_OK;
}

extern "C" int vr_op_attention(int device_id, const void* q, int32_t ldq, const void* k, int32_t ldk, const void* v,
                               int32_t ldv, void* out, int32_t ldo, const int32_t* cu_q, const int32_t* cu_kv, int32_t B,
                               int32_t heads, int32_t head_dim, int32_t max_q, int32_t causal, int32_t q_shared,
                               float scale, void* stream) {
    if (!q || !k || !v || !out || !cu_q || !cu_kv) return fail(VR_ERR_INVALID, "NULL argument");
    VRCHK(set_dev(device_id));
    AttnArgs a{};
    a.q = q; a.ldq = ldq; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.out = out; a.ldo = ldo; a.cu_q = cu_q;
    a.cu_kv = cu_kv; a.B = B; a.heads = heads; a.head_dim = head_dim; a.max_q = max_q; a.causal = causal;
    a.q_shared = q_shared; a.scale = scale;
    HIPCHK(launch_attention(a, (hipStream_t)stream));
    return VR_OK;
}
