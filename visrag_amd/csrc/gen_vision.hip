// Host side of the EVisRAG generator's VISION TOWER in libvisrag_hip.so (include/visrag_gen.h: vg_vision_*).
// Reference boundary: src/evisrag/predict.py:98-103,140,147 — up to five page images per prompt go to vLLM with the
// prompt; vLLM's Qwen2.5-VL runs this tower on the processor's pixel rows and puts the result at the <|image_pad|>
// positions.  Architecture (HF modeling_qwen2_5_vl.py:99-175, 211-325, 406-470; restated on the CPU in
// oracle/qwen_vision_oracle.py): patch embedding (a Conv3d whose kernel equals its stride = one GEMM), `depth` blocks of
// RMSNorm -> attention with 2-D rotary positions -> RMSNorm -> SwiGLU MLP (all linears with bias), attention inside
// 112-pixel windows except for the `fullatt` blocks that see the whole image, then the 2 x 2 patch merger (RMSNorm,
// GELU MLP) into the language model's hidden size.
//
// How it maps onto this library's kernels (no kernel of its own beyond two row movers):
//   * rows are put into WINDOW ORDER once, while the pixel rows are converted to bf16 (gather_rows_bf16), so every
//     attention segment — a window or an image — is a contiguous row range handed to the flash-attention kernel as
//     cu_q / cu_kv; the merger's output is scattered back into image order straight into the prefill's embedding rows.
//   * head_dim 80 (every released Qwen2.5-VL tower: 1280 / 16) runs on the attention kernel's own head_dim-80 form.  Any
//     other head_dim (a multiple of 4 up to 128; the test fixture's 40) gets a 128-wide slot per head: the q / k / v weight rows of a head are
//     packed as two halves at slot rows [0, hd/2) and [64, 64 + hd/2) (zeros elsewhere) and the output projection's
//     columns likewise, so rotate-half pairs (c, c + hd/2) land on slot pairs (p, p + 64), the padded channels are
//     exact zeros through q.k and p.v, and the softmax scale stays 1/sqrt(hd).
//   * rotary: the (cos, sin) of every row and slot pair is the same for all blocks and heads — computed once per call
//     (rope2d_table_kernel), then each block rotates its q and k heads in place in the qkv rows and the attention kernel
//     reads q, k and v straight from that buffer (one row stride for all three): no q / k / v copies.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "gen_model.h"

namespace {

struct VisBlock {
    Vec n1, n2;
    Linear qkv, proj, gu, down;
    int parts_gu_w = 0, parts_gu_b = 0;
};

// token geometry of one vg_vision_encode call (host)
struct VisPlan {
    int rows = 0, tokens = 0, max_win = 0, max_img = 0;
    std::vector<int> order;        // [tokens] merged token at window-order place i
    std::vector<int> win_bounds;   // row boundaries of the windows (window order)
    std::vector<int> img_bounds;   // row boundaries of the frames
    std::vector<int> hw;           // [rows][2] patch coordinates in the processor's (merge-block major) row order
};

}  // namespace

struct VisionTower {
    vg_vision_config_t c{};
    int H = 0, Hp = 0, heads = 0, hd = 0, half = 0, hh = 0, AD = 0, QKVp = 0, ATp = 0, I = 0, Ip = 0, PD = 0, PDp = 0, m2 = 0, MH = 0, MHp = 0;
    // (hh: distance of a rotate-half pair inside a head slot = half the slot width; AD = heads x slot width)
    int Rcap = 0;
    std::vector<bool> full;
    Linear patch;
    std::vector<VisBlock> blocks;
    Vec ln_q;
    Linear mlp0, mlp2;
    DevBuf inv_freq;
    DevBuf w_pix, w_px, w_x, w_xn, w_qkv, w_cs, w_att, w_act, w_mid, w_out, w_perm, w_pos, w_cuw, w_cui, w_order;
    DevBuf w_rowimg, w_pageptr;      // page input: image of every window-ordered row; page pointers + widths
};

// The window order of HF's get_vision_window_index (transformers/vision_utils.py:130-188) and the (h, w) coordinates of
// get_vision_position_ids (:81-127), from the grids alone.  A window is window_size / patch / merge merged tokens on a
// side; windows at the right / bottom edge are smaller; inside a window tokens run row-major.
static int vision_plan(const vg_vision_config_t& c, const int32_t* grid_thw, int n_images, VisPlan& p) {
    const int m = c.spatial_merge_size, ws = c.window_size / m / c.patch_size;
    if (ws <= 0) return fail(VR_ERR_INVALID, "window_size %d is smaller than one merged token", c.window_size);
    p = VisPlan();
    p.win_bounds.push_back(0);
    p.img_bounds.push_back(0);
    int base = 0;
    for (int i = 0; i < n_images; ++i) {
        const int t = grid_thw[3 * i], gh = grid_thw[3 * i + 1], gw = grid_thw[3 * i + 2];
        if (t <= 0 || gh <= 0 || gw <= 0 || gh % m || gw % m) return fail(VR_ERR_INVALID, "image %d: bad grid %d x %d x %d", i, t, gh, gw);
        const int lh = gh / m, lw = gw / m;
        for (int f = 0; f < t; ++f) {
            for (int wy = 0; wy * ws < lh; ++wy)
                for (int wx = 0; wx * ws < lw; ++wx) {
                    int n = 0;
                    for (int y = wy * ws; y < std::min(lh, (wy + 1) * ws); ++y)
                        for (int x = wx * ws; x < std::min(lw, (wx + 1) * ws); ++x, ++n) p.order.push_back(base + (f * lh + y) * lw + x);
                    p.win_bounds.push_back(p.win_bounds.back() + n * m * m);
                    p.max_win = std::max(p.max_win, n * m * m);
                }
            p.img_bounds.push_back(p.img_bounds.back() + gh * gw);
            for (int by = 0; by < lh; ++by)
                for (int bx = 0; bx < lw; ++bx)
                    for (int iy = 0; iy < m; ++iy)
                        for (int ix = 0; ix < m; ++ix) { p.hw.push_back(by * m + iy); p.hw.push_back(bx * m + ix); }
        }
        p.max_img = std::max(p.max_img, gh * gw);
        base += t * lh * lw;
    }
    p.tokens = base;
    p.rows = base * m * m;
    return VR_OK;
}

static int check_vision_config(const vg_vision_config_t& c) {
    if (c.depth <= 0 || c.hidden_size <= 0 || c.num_heads <= 0 || c.hidden_size % c.num_heads) return fail(VR_ERR_INVALID, "bad tower size");
    const int hd = c.hidden_size / c.num_heads;
    if (hd % 4 || hd > 128) return fail(VR_ERR_INVALID, "vision head_dim %d: must be a multiple of 4 and at most 128", hd);
    if (c.patch_size <= 0 || c.temporal_patch_size <= 0 || c.in_channels <= 0 || c.spatial_merge_size <= 0 || c.intermediate_size <= 0)
        return fail(VR_ERR_INVALID, "bad patch / merge / MLP size");
    if (c.n_fullatt < 0 || c.n_fullatt > 16) return fail(VR_ERR_INVALID, "n_fullatt must be in [0, 16]");
    if (c.hidden_size > 3584) return fail(VR_ERR_INVALID, "vision hidden size above 3584");
    return VR_OK;
}

extern "C" int vg_vision_plan(const vg_vision_config_t* cfg, const int32_t* grid_thw, int32_t n_images, int32_t* order,
                              int32_t* win_bounds, int32_t* n_windows, int32_t* hw) {
    if (!cfg || !grid_thw || n_images <= 0) return fail(VR_ERR_INVALID, "NULL / empty argument");
    VRCHK(check_vision_config(*cfg));
    VisPlan p;
    VRCHK(vision_plan(*cfg, grid_thw, n_images, p));
    if (order) std::copy(p.order.begin(), p.order.end(), order);
    if (win_bounds) std::copy(p.win_bounds.begin(), p.win_bounds.end(), win_bounds);
    if (n_windows) *n_windows = (int)p.win_bounds.size() - 1;
    if (hw) std::copy(p.hw.begin(), p.hw.end(), hw);
    return VR_OK;
}

static int alloc_linear(Linear& L, int n_total, int k) {
    if (L.w.p) return (L.n == n_total && L.k == k) ? VR_OK : fail(VR_ERR_INVALID, "inconsistent shapes for a packed weight");
    L.n = n_total; L.k = k; L.n_pad = pad128(n_total); L.k_pad = pad128(k);
    return L.w.alloc((size_t)pad256(n_total) * L.k_pad * 2);
}
static int alloc_bias(Linear& L, int n_total) {     // (a bias may arrive before its weight)
    if (L.b.p) return VR_OK;
    return L.b.alloc((size_t)pad256(n_total) * 4);
}

extern "C" int vg_vision_create(vg_model_t m, const vg_vision_config_t* cfg) {
    if (!m || !cfg) return fail(VR_ERR_INVALID, "NULL argument");
    if (m->vis) return fail(VR_ERR_STATE, "the model already has a vision tower");
    const vg_vision_config_t& c = *cfg;
    VRCHK(check_vision_config(c));
    if (c.out_hidden_size != m->E) return fail(VR_ERR_INVALID, "out_hidden_size %d != the language model's hidden size %d", c.out_hidden_size, m->E);
    const int m2 = c.spatial_merge_size * c.spatial_merge_size;
    if (c.max_rows <= 0 || c.max_rows % m2) return fail(VR_ERR_INVALID, "max_rows must be a positive multiple of %d", m2);
    if (c.max_rows / m2 > m->Tcap) return fail(VR_ERR_CAPACITY, "max_rows / %d exceeds the prefill capacity %d", m2, m->Tcap);
    for (int i = 0; i < c.n_fullatt; ++i)
        if (c.fullatt_blocks[i] < 0 || c.fullatt_blocks[i] >= c.depth) return fail(VR_ERR_INVALID, "fullatt block %d out of range", c.fullatt_blocks[i]);
    VRCHK(set_dev(m->device));
    VisionTower* v = new VisionTower();
    m->vis = v;
    m->finalized = false;
    auto bail = [&](int rc) { vision_destroy(m); return rc; };
    v->c = c;
    v->H = c.hidden_size; v->Hp = pad128(v->H); v->heads = c.num_heads; v->hd = v->H / v->heads; v->half = v->hd / 2;
    v->hh = v->hd == 80 ? v->half : 64;      // the 7B / 3B / 72B towers' head_dim on its own attention form; else a 128-wide slot
    v->AD = v->heads * 2 * v->hh;
    v->QKVp = pad128(3 * v->AD); v->ATp = pad128(v->AD);                       // row strides of the qkv / attention-output rows
    v->I = c.intermediate_size; v->Ip = pad128(v->I);
    v->PD = c.in_channels * c.temporal_patch_size * c.patch_size * c.patch_size; v->PDp = pad128(v->PD);
    v->m2 = m2; v->MH = m2 * v->H; v->MHp = pad128(v->MH);
    v->blocks.resize(c.depth);
    v->full.assign(c.depth, false);
    for (int i = 0; i < c.n_fullatt; ++i) v->full[c.fullatt_blocks[i]] = true;
    // rotary frequencies: lane p < 64 rotates slot channels (p, p + 64) by pos * table[p].  Tower:
    // channel pair j < hd/4 turns with the row's h coordinate, hd/4 <= j < hd/2 with w, both with frequency
    // 10000^(-2 (j mod hd/4) / (hd/2))  (modeling_qwen2_5_vl.py:129-141, 441-446); slots past hd/2 hold zeros.
    {
        const int quarter = v->hd / 4;
        std::vector<float> tab(64, 0.f);
        for (int p = 0; p < v->half && p < 64; ++p) tab[p] = 1.0f / powf(10000.0f, (float)(2 * (p % quarter)) / (float)v->half);
        int rc = v->inv_freq.alloc(64 * 4);
        if (rc != VR_OK) return bail(rc);
        if (hipMemcpy(v->inv_freq.p, tab.data(), 64 * 4, hipMemcpyHostToDevice) != hipSuccess) return bail(fail(VR_ERR_HIP, "hipMemcpy failed"));
    }
    const size_t R = (size_t)pad256(c.max_rows), NT = (size_t)pad256(c.max_rows / m2);
    v->Rcap = (int)R;
    struct { DevBuf* b; size_t bytes; } ws[] = {
        {&v->w_pix, (size_t)c.max_rows * v->PD * 4}, {&v->w_px, R * v->PDp * 2}, {&v->w_x, R * v->Hp * 4},
        {&v->w_xn, std::max(R, NT * m2) * v->Hp * 2},      // also read as [pad256(tokens)][m2 * Hp] by the merger
        {&v->w_qkv, R * v->QKVp * 2}, {&v->w_cs, R * 64 * 8},
        {&v->w_att, R * v->ATp * 2}, {&v->w_act, R * v->Ip * 2}, {&v->w_mid, NT * v->MHp * 2}, {&v->w_out, NT * (size_t)m->E * 4},
        {&v->w_perm, R * 4}, {&v->w_pos, 2 * R * 4}, {&v->w_cuw, (R / m2 + 2) * 4}, {&v->w_cui, (R + 2) * 4}, {&v->w_order, NT * 4}};
    for (auto& w : ws) {
        int rc = w.b->alloc(w.bytes);
        if (rc != VR_OK) return bail(rc);
    }
    return VR_OK;
}

void vision_destroy(vg_model_s* m) {
    if (m && m->vis) { delete m->vis; m->vis = nullptr; }
}

int vision_load_weight(vg_model_s* m, const std::string& key, const void* src, int bf, const int64_t* shape, int ndim, size_t numel) {
    VisionTower* v = m->vis;
    const int H = v->H, I = v->I, half = v->half, hh = v->hh, AD = v->AD;
    auto bad_shape = [&]() { return fail(VR_ERR_INVALID, "unexpected shape for visual.%s", key.c_str()); };
    auto vec = [&](Vec& x, int n, int n_alloc) { if (numel != (size_t)n) return bad_shape(); return load_vec(x, src, bf, n, n_alloc); };
    // a whole [n][k] matrix under row / column block maps, and its bias under the row map
    auto mat = [&](Linear& L, int n_total, int k_total, int rows, int cols, int rblk, int rstride, int roff, int cblk, int cstride) {
        if (!shape_is(shape, ndim, {rows, cols})) return bad_shape();
        VRCHK(alloc_linear(L, n_total, k_total));
        HIPCHK(launch_pack_weight_blocks(src, bf, rows, cols, cols, (char*)L.w.p + (size_t)roff * L.k_pad * 2, L.k_pad, rblk, rstride, cblk, cstride, 0));
        HIPCHK(hipDeviceSynchronize());
        L.has_w = true;
        return (int)VR_OK;
    };
    auto bias = [&](Linear& L, int n_total, int n, int blk, int stride, int off) {
        if (numel != (size_t)n) return bad_shape();
        VRCHK(alloc_bias(L, n_total));
        HIPCHK(launch_to_f32_blocks(src, bf, L.b.as<float>(), (size_t)n, blk, stride, off, 0));
        HIPCHK(hipDeviceSynchronize());
        L.has_b = true;
        return (int)VR_OK;
    };
    if (key == "patch_embed.proj.weight") {
        if (numel != (size_t)H * v->PD || ndim < 2 || shape[0] != H) return bad_shape();
        VRCHK(alloc_linear(v->patch, H, v->PD));
        HIPCHK(launch_pack_weight_blocks(src, bf, H, v->PD, v->PD, v->patch.w.p, v->patch.k_pad, H, H, v->PD, v->PD, 0));
        HIPCHK(hipDeviceSynchronize());
        v->patch.has_w = true;
        return VR_OK;
    }
    if (key == "merger.ln_q.weight") return vec(v->ln_q, H, v->Hp);
    // merger.mlp.0 reads the m2 rows of a merged token side by side: columns u * H + c sit at u * Hp + c of the normed rows
    if (key == "merger.mlp.0.weight") return mat(v->mlp0, v->MH, v->m2 * v->Hp, v->MH, v->MH, v->MH, v->MH, 0, H, v->Hp);
    if (key == "merger.mlp.0.bias") return bias(v->mlp0, v->MH, v->MH, v->MH, v->MH, 0);
    if (key == "merger.mlp.2.weight") return mat(v->mlp2, m->E, v->MH, m->E, v->MH, m->E, m->E, 0, v->MH, v->MH);
    if (key == "merger.mlp.2.bias") return bias(v->mlp2, m->E, m->E, m->E, m->E, 0);
    if (key.rfind("blocks.", 0) == 0) {
        int n = -1, off = 0;
        if (sscanf(key.c_str(), "blocks.%d.%n", &n, &off) < 1) return fail(VR_ERR_INVALID, "bad key visual.%s", key.c_str());
        if (n < 0 || n >= v->c.depth) return fail(VR_ERR_INVALID, "vision block %d out of range", n);
        const std::string sub = key.substr(off);
        VisBlock& b = v->blocks[n];
        if (sub == "norm1.weight") return vec(b.n1, H, v->Hp);
        if (sub == "norm2.weight") return vec(b.n2, H, v->Hp);
        // q | k | v rows of head h, channel c: slot row (part * heads + h) * 2 hh + (c < half ? c : hh + c - half)
        if (sub == "attn.qkv.weight") return mat(b.qkv, 3 * AD, H, 3 * H, H, half, hh, 0, H, H);
        if (sub == "attn.qkv.bias") return bias(b.qkv, 3 * AD, 3 * H, half, hh, 0);
        if (sub == "attn.proj.weight") return mat(b.proj, H, AD, H, H, H, H, 0, half, hh);
        if (sub == "attn.proj.bias") return bias(b.proj, H, H, H, H, 0);
        // 16-row [gate | up] interleave of EPI_SWIGLU, bias likewise
        for (int up = 0; up < 2; ++up) {
            const std::string nm = up ? "mlp.up_proj." : "mlp.gate_proj.";
            if (sub == nm + "weight") { b.parts_gu_w |= 1 << up; return mat(b.gu, 2 * v->Ip, H, I, H, 16, 32, up * 16, H, H); }
            if (sub == nm + "bias") { b.parts_gu_b |= 1 << up; return bias(b.gu, 2 * v->Ip, I, 16, 32, up * 16); }
        }
        if (sub == "mlp.down_proj.weight") return mat(b.down, H, I, H, I, H, H, 0, I, I);
        if (sub == "mlp.down_proj.bias") return bias(b.down, H, H, H, H, 0);
    }
    return fail(VR_ERR_INVALID, "unknown key visual.%s", key.c_str());
}

int vision_check_complete(const vg_model_s* m) {
    const VisionTower* v = m->vis;
    auto full = [](const Linear& L) { return L.has_w && L.has_b; };
    if (!v->patch.has_w || !v->ln_q.ok || !full(v->mlp0) || !full(v->mlp2)) return fail(VR_ERR_STATE, "vision patch_embed / merger incomplete");
    for (size_t i = 0; i < v->blocks.size(); ++i) {
        const VisBlock& b = v->blocks[i];
        if (!b.n1.ok || !b.n2.ok || !full(b.qkv) || !full(b.proj) || b.parts_gu_w != 3 || b.parts_gu_b != 3 || !full(b.down))
            return fail(VR_ERR_STATE, "vision block %zu is incomplete", i);
    }
    return VR_OK;
}

// pixels != NULL: the processor's f32 pixel rows (host); else `pages`: n_images u8 RGB (HWC) images of exactly
// (grid_h * patch) x (grid_w * patch) pixels, all on the host or all on the device, normalised with (mean, std) here
static int vision_encode_impl(vg_model_t m, const float* pixels, const uint8_t* const* pages, int pages_on_device,
                              const float* mean3, const float* std3, const int32_t* grid_thw, int32_t n_images,
                              float* embeds_out, void* stream);

extern "C" int vg_vision_encode(vg_model_t m, const float* pixels, const int32_t* grid_thw, int32_t n_images, float* embeds_out,
                                void* stream) {
    if (!pixels) return fail(VR_ERR_INVALID, "NULL argument");
    return vision_encode_impl(m, pixels, nullptr, 0, nullptr, nullptr, grid_thw, n_images, embeds_out, stream);
}

extern "C" int vg_vision_encode_pages(vg_model_t m, const uint8_t* const* pages, int32_t pages_on_device, const float* mean3,
                                      const float* std3, const int32_t* grid_thw, int32_t n_images, float* embeds_out,
                                      void* stream) {
    if (!pages || !mean3 || !std3) return fail(VR_ERR_INVALID, "NULL argument");
    for (int i = 0; i < n_images; ++i)
        if (!pages[i]) return fail(VR_ERR_INVALID, "page %d is NULL", i);
    for (int c = 0; c < 3; ++c)
        if (!(std3[c] > 0.f)) return fail(VR_ERR_INVALID, "image_std must be positive");
    return vision_encode_impl(m, nullptr, pages, pages_on_device, mean3, std3, grid_thw, n_images, embeds_out, stream);
}

static int vision_encode_impl(vg_model_t m, const float* pixels, const uint8_t* const* pages, int pages_on_device,
                              const float* mean3, const float* std3, const int32_t* grid_thw, int32_t n_images,
                              float* embeds_out, void* stream) {
    if (!m || !grid_thw) return fail(VR_ERR_INVALID, "NULL argument");
    if (!m->vis) return fail(VR_ERR_STATE, "no vision tower (vg_vision_create)");
    if (!m->finalized) return fail(VR_ERR_STATE, "vg_finalize has not succeeded");
    if (m->running) return fail(VR_ERR_STATE, "a free run is in progress (vg_run_end first)");
    if (n_images <= 0) return fail(VR_ERR_INVALID, "no images");
    VisionTower* v = m->vis;
    VisPlan p;
    VRCHK(vision_plan(v->c, grid_thw, n_images, p));
    if (p.rows > v->c.max_rows) return fail(VR_ERR_CAPACITY, "%d patch rows (max_rows %d)", p.rows, v->c.max_rows);
    VRCHK(set_dev(m->device));
    hipStream_t s = (hipStream_t)stream;
    const int R = p.rows, NT = p.tokens, m2 = v->m2, H = v->H, Hp = v->Hp, AD = v->AD, E = m->E;
    m->vis_tokens = 0;
    // ---- geometry to the device: row permutation, (h, w) of every permuted row, segment boundaries
    std::vector<int> perm(R), pos(2 * (size_t)R);
    for (int i = 0; i < NT; ++i)
        for (int u = 0; u < m2; ++u) perm[i * m2 + u] = p.order[i] * m2 + u;
    for (int r = 0; r < R; ++r) { pos[r] = p.hw[2 * perm[r]]; pos[(size_t)R + r] = p.hw[2 * perm[r] + 1]; }
    HIPCHK(hipMemcpyAsync(v->w_perm.p, perm.data(), (size_t)R * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(v->w_pos.p, pos.data(), pos.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(v->w_cuw.p, p.win_bounds.data(), p.win_bounds.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(v->w_cui.p, p.img_bounds.data(), p.img_bounds.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(v->w_order.p, p.order.data(), (size_t)NT * 4, hipMemcpyHostToDevice, s));
    std::vector<int> row_img;                         // pages path: image of every permuted row
    std::vector<const uint8_t*> page_dev;
    std::vector<int> page_w;
    if (pixels) {
        HIPCHK(hipMemcpyAsync(v->w_pix.p, pixels, (size_t)R * v->PD * 4, hipMemcpyHostToDevice, s));
    } else {
        if (v->c.in_channels != 3) return fail(VR_ERR_INVALID, "page input needs a 3-channel tower");
        // rows of image i in processor order: [img_bounds[i], img_bounds[i+1]) (still images: one frame each)
        row_img.resize(R);
        std::vector<int> owner(R);
        for (int i = 0, r = 0; i < n_images; ++i) {
            if (grid_thw[3 * i] != 1) return fail(VR_ERR_INVALID, "page input is for still images (t = 1)");
            for (int k = 0; k < grid_thw[3 * i + 1] * grid_thw[3 * i + 2]; ++k) owner[r++] = i;
        }
        for (int r = 0; r < R; ++r) row_img[r] = owner[perm[r]];
        page_dev.resize(n_images); page_w.resize(n_images);
        size_t total = 0;
        std::vector<size_t> off(n_images);
        const int P = v->c.patch_size;
        for (int i = 0; i < n_images; ++i) {
            page_w[i] = grid_thw[3 * i + 2] * P;
            off[i] = total;
            total += ((size_t)grid_thw[3 * i + 1] * P * page_w[i] * 3 + 15) / 16 * 16;
        }
        if (pages_on_device) {
            for (int i = 0; i < n_images; ++i) page_dev[i] = pages[i];
        } else {
            // (w_pix holds max_rows f32 pixel rows: 4 bytes per u8 the pages can have)
            if (total > v->w_pix.bytes) return fail(VR_ERR_CAPACITY, "pages exceed the pixel staging buffer");
            for (int i = 0; i < n_images; ++i) {
                const size_t nb = (size_t)grid_thw[3 * i + 1] * P * page_w[i] * 3;
                HIPCHK(hipMemcpyAsync((char*)v->w_pix.p + off[i], pages[i], nb, hipMemcpyHostToDevice, s));
                page_dev[i] = (const uint8_t*)v->w_pix.p + off[i];
            }
        }
        VRCHK(v->w_rowimg.reserve((size_t)R * 4));
        VRCHK(v->w_pageptr.reserve((size_t)n_images * 8 + (size_t)n_images * 4 + 32));
        HIPCHK(hipMemcpyAsync(v->w_rowimg.p, row_img.data(), (size_t)R * 4, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(v->w_pageptr.p, page_dev.data(), (size_t)n_images * 8, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync((char*)v->w_pageptr.p + (size_t)n_images * 8, page_w.data(), (size_t)n_images * 4, hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipStreamSynchronize(s));                  // the host vectors above go away with this frame
    HIPCHK(launch_rope2d_table(v->w_pos.as<int>(), v->w_pos.as<int>() + R, R, v->hd / 4, v->inv_freq.as<float>(), v->w_cs.p, s));
    // ---- patch embedding over the permuted bf16 pixel rows
    if (pixels) {
        HIPCHK(launch_gather_rows_bf16(v->w_pix.as<float>(), v->w_perm.as<int>(), R, v->PD, v->w_px.p, v->PDp, s));
    } else {
        // u8 page -> /255, normalise -> bf16 patch row, straight into window order (what the processor + the row mover do)
        HIPCHK(launch_patch_rows_u8((const uint8_t* const*)v->w_pageptr.p, (const int*)((const char*)v->w_pageptr.p + (size_t)n_images * 8),
                                    v->w_rowimg.as<int>(), v->w_pos.as<int>(), v->w_pos.as<int>() + R, R, v->c.patch_size,
                                    v->c.temporal_patch_size, mean3, std3, v->w_px.p, v->PDp, s));
    }
    float* x = v->w_x.as<float>();
    { GemmArgs a = gen_gemm_args(v->w_px.p, v->PDp, v->patch, R, x, Hp); HIPCHK(launch_gemm(a, EPI_F32, GEMM_VARIANT_AUTO, s)); }
    const float eps = v->c.rms_norm_eps;
    HIPCHK(launch_rmsnorm(x, R, H, Hp, v->blocks[0].n1.v.as<float>(), eps, v->w_xn.p, Hp, s));
    const int nb = (int)v->blocks.size();
    for (int l = 0; l < nb; ++l) {
        VisBlock& b = v->blocks[l];
        { GemmArgs a = gen_gemm_args(v->w_xn.p, Hp, b.qkv, R, v->w_qkv.p, v->QKVp); HIPCHK(launch_gemm(a, EPI_BF16, GEMM_VARIANT_AUTO, s)); }
        HIPCHK(launch_rope2d_inplace(v->w_qkv.p, v->QKVp, R, 2 * v->heads, v->hh, v->w_cs.p, s));
        {
            AttnArgs a{};
            const char* qkv = (const char*)v->w_qkv.p;
            a.q = qkv; a.k = qkv + (size_t)AD * 2; a.v = qkv + (size_t)AD * 4; a.ldq = a.ldk = a.ldv = v->QKVp;
            a.out = v->w_att.p; a.ldo = v->ATp;
            const bool fullatt = v->full[l];
            a.cu_q = a.cu_kv = fullatt ? v->w_cui.as<int>() : v->w_cuw.as<int>();
            a.B = (int)(fullatt ? p.img_bounds.size() : p.win_bounds.size()) - 1;
            a.max_q = fullatt ? p.max_img : p.max_win;
            a.heads = v->heads; a.head_dim = 2 * v->hh; a.scale = 1.0f / sqrtf((float)v->hd); a.causal = 0; a.q_shared = 0; a.kv_group = 1;
            HIPCHK(launch_attention(a, s));
        }
        { GemmArgs a = gen_gemm_args(v->w_att.p, v->ATp, b.proj, R, x, Hp); a.resid = x; HIPCHK(launch_gemm(a, EPI_RESID, GEMM_VARIANT_AUTO, s)); }
        HIPCHK(launch_rmsnorm(x, R, H, Hp, b.n2.v.as<float>(), eps, v->w_xn.p, Hp, s));
        { GemmArgs a = gen_gemm_args(v->w_xn.p, Hp, b.gu, R, v->w_act.p, v->Ip); HIPCHK(launch_gemm(a, EPI_SWIGLU, GEMM_VARIANT_AUTO, s)); }
        { GemmArgs a = gen_gemm_args(v->w_act.p, v->Ip, b.down, R, x, Hp); a.resid = x; HIPCHK(launch_gemm(a, EPI_RESID, GEMM_VARIANT_AUTO, s)); }
        const float* next = l + 1 < nb ? v->blocks[l + 1].n1.v.as<float>() : v->ln_q.v.as<float>();
        HIPCHK(launch_rmsnorm(x, R, H, Hp, next, eps, v->w_xn.p, Hp, s));
    }
    // ---- merger: the m2 normed rows of a merged token side by side -> GELU MLP -> fp32 rows, back in image order
    { GemmArgs a = gen_gemm_args(v->w_xn.p, m2 * Hp, v->mlp0, NT, v->w_mid.p, v->MHp); HIPCHK(launch_gemm(a, EPI_GELU, GEMM_VARIANT_AUTO, s)); }
    { GemmArgs a = gen_gemm_args(v->w_mid.p, v->MHp, v->mlp2, NT, v->w_out.p, E); HIPCHK(launch_gemm(a, EPI_F32, GEMM_VARIANT_AUTO, s)); }
    HIPCHK(launch_scatter_rows(v->w_out.as<float>(), v->w_order.as<int>(), NT, E, m->w_emb.as<float>(), E, s));
    if (embeds_out) HIPCHK(hipMemcpyAsync(embeds_out, m->w_emb.p, (size_t)NT * E * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    m->vis_tokens = NT;
    return VR_OK;
}
