"""The vision-tower oracle (oracle/qwen_vision_oracle.py) against the fixtures the HuggingFace Qwen2.5-VL
implementation produced in the build container (oracle/gen_golden_evisrag_vision.py ->
tests/golden/evisrag_vision_tiny.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle.qwen_gen_oracle import QwenGenOracle, synth_weights, tiny_config
from oracle.qwen_vision_oracle import (QwenVisionOracle, hd80_vision_config, image_bounds, patchify, position_hw, smart_resize,
                                       tiny_vision_config, vision_weight_specs, window_order)
from visrag_amd.evisrag import rope_index

GOLD = os.path.join(os.path.dirname(__file__), "golden", "evisrag_vision_tiny.npz")


def _weights(g):
    return {k[2:]: torch.from_numpy(g[k]).view(torch.bfloat16).float() for k in g.files if k.startswith("w:")}


def _pixels(g):
    return torch.from_numpy(g["pixels_bf16"]).view(torch.bfloat16).float()


def _grids(g):
    return [tuple(int(v) for v in r) for r in g["grids"]]


def test_fixture_holds_every_tower_tensor():
    g = np.load(GOLD)
    w = _weights(g)
    for name, shape, _ in vision_weight_specs(tiny_vision_config()):
        assert tuple(w[name].shape) == shape, name


def test_tower_embeddings_match_hf():
    g = np.load(GOLD)
    o = QwenVisionOracle(tiny_vision_config(), _weights(g))
    emb, rows = o.forward(_pixels(g), _grids(g), return_rows=True)
    np.testing.assert_allclose(emb.numpy(), g["image_embeds"], rtol=2e-4, atol=2e-5)
    # HF's per-row hidden state is in window order; the oracle's return_rows is in patchify order
    order, _ = window_order(_grids(g), o.cfg)
    perm = (order[:, None] * 4 + torch.arange(4)[None]).reshape(-1)
    np.testing.assert_allclose(rows[perm].numpy(), g["tower_rows_window_order"], rtol=2e-4, atol=2e-5)


def test_hd80_tower_embeddings_match_hf():
    g = np.load(os.path.join(os.path.dirname(GOLD), "evisrag_vision_hd80.npz"))
    o = QwenVisionOracle(hd80_vision_config(), _weights(g))
    emb = o.forward(_pixels(g), _grids(g))
    np.testing.assert_allclose(emb.numpy(), g["image_embeds"], rtol=2e-4, atol=2e-5)


def test_window_order_geometry():
    cfg = tiny_vision_config()               # windows of 2 x 2 merged tokens
    order, bounds = window_order([(1, 10, 6)], cfg)      # 5 x 3 merged tokens
    assert order.tolist() == [0, 1, 3, 4, 2, 5, 6, 7, 9, 10, 8, 11, 12, 13, 14]
    assert bounds == [0, 16, 24, 40, 48, 56, 60]
    order, bounds = window_order([(1, 4, 8)], cfg)       # exact multiple: no empty windows survive
    assert order.tolist() == [0, 1, 4, 5, 2, 3, 6, 7] and bounds == [0, 16, 32]
    order, bounds = window_order([(1, 4, 4), (1, 2, 2)], cfg)
    assert order.tolist() == [0, 1, 2, 3, 4] and bounds == [0, 16, 20]
    assert image_bounds([(1, 4, 4), (2, 2, 2)]) == [0, 16, 20, 24]
    hw = position_hw([(1, 4, 4)], 2)
    assert hw[:6].tolist() == [[0, 0], [0, 1], [1, 0], [1, 1], [0, 2], [0, 3]]


def test_smart_resize_and_patchify_match_the_hf_processor():
    from PIL import Image
    g = np.load(GOLD)
    rgb = g["proc_rgb"]
    mn, mx = (int(v) for v in g["proc_min_max_pixels"])
    H, W = smart_resize(rgb.shape[0], rgb.shape[1], 28, mn, mx)
    _, gh, gw = (int(v) for v in g["proc_grid"][0])
    assert (H, W) == (gh * 14, gw * 14)
    img = np.asarray(Image.fromarray(rgb).resize((W, H), Image.BICUBIC)).astype(np.float32)
    x = (img / 255.0 - g["proc_mean"]) / g["proc_std"]
    px, grid = patchify(torch.from_numpy(x).permute(2, 0, 1), tiny_vision_config())
    assert grid == (1, gh, gw)
    np.testing.assert_allclose(px.numpy(), g["proc_pixel_values"], rtol=1e-5, atol=1e-5)


# expected values: HF's smart_resize (image_processing_pil_qwen2_vl) run in the build container
@pytest.mark.parametrize("hw,want", [((1000, 700), (1008, 700)), ((3000, 2000), (1204, 812)), ((20, 30), (56, 84)),
                                     ((28, 28), (56, 56)), ((1, 150), (28, 700)), ((4000, 100), (4004, 112))])
def test_smart_resize_cases(hw, want):
    assert smart_resize(*hw, 28, 3136, 1003520) == want
    with pytest.raises(ValueError):
        smart_resize(1, 201)


def test_prompt_with_images_matches_hf_logits_and_positions():
    """text + three images through tower and language model: the placeholder rows take the tower's embeddings, the
    positions are the product's rope_index (pinned here to HF's get_rope_index)."""
    g = np.load(GOLD)
    cfg = tiny_config()
    vo = QwenVisionOracle(tiny_vision_config(cfg.hidden_size), _weights(g))
    lm = QwenGenOracle(cfg, synth_weights(cfg, seed=int(g["lm_seed"])))
    ids = torch.from_numpy(g["prompt_ids"]).long()
    grids = _grids(g)
    pos = rope_index(ids.tolist(), int(g["image_token_id"]), [(h // 2, w // 2) for _, h, w in grids])
    np.testing.assert_array_equal(pos, g["prompt_pos3"])
    emb = lm.embed(ids).clone()
    emb[ids == int(g["image_token_id"])] = vo.forward(_pixels(g), grids)
    lm.reset()
    logits = lm.forward(emb, torch.from_numpy(pos).long())[-1]
    np.testing.assert_allclose(logits.numpy(), g["prompt_logits"], rtol=2e-4, atol=3e-5)


# ---------------------------------------------------------------------------------------------------------------
# the product's host side (visrag_amd/evisrag.py + the host-only entry point of the library) against the oracle
# ---------------------------------------------------------------------------------------------------------------

def _product_vc(cfg):
    from visrag_amd.evisrag import VisionConfig
    return VisionConfig(depth=cfg.depth, hidden_size=cfg.hidden_size, num_heads=cfg.num_heads, intermediate_size=cfg.intermediate_size,
                        out_hidden_size=cfg.out_hidden_size, window_size=cfg.window_size,
                        fullatt_block_indexes=tuple(cfg.fullatt_block_indexes))


@pytest.mark.parametrize("grids", [[(1, 10, 6), (1, 4, 8), (1, 2, 2)], [(2, 6, 10)], [(1, 32, 32), (1, 18, 46)], [(1, 2, 2)]])
@pytest.mark.parametrize("full_size", [False, True])
def test_library_token_geometry_matches_the_oracle(grids, full_size):
    """vg_vision_plan (host-only C++ in gen_vision.hip) == the oracle's window order / boundaries / coordinates."""
    from oracle.qwen_vision_oracle import QwenVisionConfig
    from visrag_amd.evisrag import vision_plan
    cfg = QwenVisionConfig() if full_size else tiny_vision_config()        # windows of 4 x 4 or 2 x 2 merged tokens
    order, bounds, hw = vision_plan(_product_vc(cfg), grids)
    o_order, o_bounds = window_order(grids, cfg)
    assert order.tolist() == o_order.tolist()
    assert bounds.tolist() == o_bounds
    np.testing.assert_array_equal(hw, position_hw(grids, cfg.spatial_merge_size).numpy())


def test_process_images_matches_the_hf_processor():
    from PIL import Image
    from visrag_amd.evisrag import VisionConfig, process_images, smart_resize as product_smart_resize
    g = np.load(GOLD)
    mn, mx = (int(v) for v in g["proc_min_max_pixels"])
    vc = VisionConfig(min_pixels=mn, max_pixels=mx)
    px, grid = process_images([Image.fromarray(g["proc_rgb"])], vc)
    np.testing.assert_array_equal(grid, g["proc_grid"])
    np.testing.assert_allclose(px, g["proc_pixel_values"], rtol=1e-5, atol=1e-5)
    for hw in [(1000, 700), (3000, 2000), (20, 30), (1, 150), (4000, 100)]:
        assert product_smart_resize(*hw, 28, 3136, 1003520) == smart_resize(*hw, 28, 3136, 1003520)
    # two pages of different sizes: rows concatenate, one grid row per page
    px2, grid2 = process_images([Image.fromarray(g["proc_rgb"]), Image.fromarray(g["proc_rgb"][:60, :90])], vc)
    assert grid2.shape == (2, 3) and px2.shape[0] == int((grid2[:, 1] * grid2[:, 2]).sum())
    np.testing.assert_array_equal(px2[:px.shape[0]], px)


def test_head_slot_layout_reproduces_the_oracle():
    """The layout gen_vision.hip gives the attention side — every head in a 128-wide slot, its two rotate-half halves at
    slot channels [0, hd/2) and [64, 64 + hd/2), rotary over slot pairs (p, p + 64) with the table
    f[p] = 10000^(-2 (p mod hd/4) / (hd/2)) and the position component h for p < hd/4, w beyond — is arithmetic-for-
    arithmetic the tower's attention.  (fp32 torch emulation of the packing formulas; the kernels are -m gpu.)"""
    g = np.load(GOLD)
    cfg = tiny_vision_config()
    w = _weights(g)
    H, nh, hd = cfg.hidden_size, cfg.num_heads, cfg.head_dim
    half, quarter = hd // 2, hd // 4
    grids = _grids(g)
    torch.manual_seed(0)
    x = torch.randn(96, H)
    order, wb = window_order(grids, cfg)
    perm = (order[:, None] * 4 + torch.arange(4)[None]).reshape(-1)
    hw = position_hw(grids, 2)[perm].float()
    # reference: the oracle's attention of block 0 on window-ordered rows
    o = QwenVisionOracle(cfg, w)
    cos, sin = o.rotary(grids)
    cos, sin = cos[perm], sin[perm]
    b = "model.visual.blocks.0."
    qkv = (x @ w[b + "attn.qkv.weight"].T + w[b + "attn.qkv.bias"]).reshape(96, 3, nh, hd)
    rot = lambda t: torch.cat([-t[..., hd // 2:], t[..., :hd // 2]], -1)   # noqa: E731
    q = qkv[:, 0] * cos[:, None] + rot(qkv[:, 0]) * sin[:, None]
    k = qkv[:, 1] * cos[:, None] + rot(qkv[:, 1]) * sin[:, None]
    ref = torch.empty(96, nh, hd)
    for s, e in zip(wb[:-1], wb[1:]):
        a = torch.softmax(torch.einsum("qhd,khd->hqk", q[s:e], k[s:e]) / hd ** 0.5, -1)
        ref[s:e] = torch.einsum("hqk,khd->qhd", a, qkv[s:e, 2])
    ref = ref.reshape(96, H) @ w[b + "attn.proj.weight"].T + w[b + "attn.proj.bias"]
    # emulation of the packed form
    rmap = lambda r: (r // half) * 64 + r % half                          # noqa: E731  (pack_weight_blocks: blk = half, stride 64)
    Wp = torch.zeros(3 * nh * 128, H); bp = torch.zeros(3 * nh * 128)
    idx = torch.tensor([rmap(r) for r in range(3 * H)])
    Wp[idx] = w[b + "attn.qkv.weight"]; bp[idx] = w[b + "attn.qkv.bias"]
    Pp = torch.zeros(H, nh * 128)
    Pp[:, torch.tensor([rmap(c) for c in range(H)])] = w[b + "attn.proj.weight"]
    table = torch.zeros(64)
    for p in range(half):
        table[p] = 1.0 / 10000.0 ** (2 * (p % quarter) / half)
    sel = (torch.arange(64) >= quarter).long()                            # lane < hd/4 -> h, else w
    ang = hw[:, sel] * table[None, :]                                     # [rows][64]
    qkvp = (x @ Wp.T + bp).reshape(96, 3 * nh, 128)
    x1, x2 = qkvp[..., :64], qkvp[..., 64:]
    c_, s_ = ang.cos()[:, None], ang.sin()[:, None]
    rotd = torch.cat([x1 * c_ - x2 * s_, x2 * c_ + x1 * s_], -1)
    qp, kp, vp = rotd[:, :nh], rotd[:, nh:2 * nh], qkvp[:, 2 * nh:]
    out = torch.empty(96, nh, 128)
    for s, e in zip(wb[:-1], wb[1:]):
        a = torch.softmax(torch.einsum("qhd,khd->hqk", qp[s:e], kp[s:e]) / hd ** 0.5, -1)
        out[s:e] = torch.einsum("hqk,khd->qhd", a, vp[s:e])
    got = out.reshape(96, nh * 128) @ Pp.T + w[b + "attn.proj.bias"]
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------------------------------------------------------
# LLM(model=<checkpoint directory>): the host-side reader (configs, key layout, tokenizer)
# ---------------------------------------------------------------------------------------------------------------

def test_checkpoint_directory_reader(tmp_path):
    from tests.evisrag_ckpt_util import make_tiny_checkpoint
    from visrag_amd.evisrag import gen_weight_specs, iter_checkpoint_weights, load_tokenizer, read_checkpoint_configs
    d = str(tmp_path / "ckpt")
    made = make_tiny_checkpoint(d)
    cfg, vc, tied = read_checkpoint_configs(d)
    t, v = made["cfg"], made["vcfg"]
    assert (cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size,
            cfg.vocab_size) == (t.hidden_size, t.num_hidden_layers, t.num_attention_heads, t.num_key_value_heads,
                                t.intermediate_size, t.vocab_size)
    assert tuple(cfg.mrope_section) == tuple(t.mrope_section) and cfg.rope_theta == t.rope_theta and cfg.image_token_id == 5
    assert cfg.eos_token_ids == (3, 4) and tied is False                 # generation_config.json wins over config.json
    assert (vc.depth, vc.hidden_size, vc.num_heads, vc.intermediate_size, vc.out_hidden_size, vc.window_size, vc.in_channels) == \
           (v.depth, v.hidden_size, v.num_heads, v.intermediate_size, 256, v.window_size, 3)
    assert tuple(vc.fullatt_block_indexes) == tuple(v.fullatt_block_indexes)
    assert (vc.min_pixels, vc.max_pixels) == (56 * 56, 28 * 28 * 24)
    # transformers-4.51 key names come out in the layout vg_load_weight reads, every tensor once, values intact
    got = dict(iter_checkpoint_weights(d, tied))
    want = set(gen_weight_specs(cfg)) | {k for k in made["weights"] if k.startswith("model.visual.")}
    assert set(got) == want
    k = "model.language_model.layers.1.mlp.down_proj.weight"
    assert torch.equal(got[k], made["weights"][k])
    # the nested layout of later transformers versions reads the same
    import json
    j = json.load(open(os.path.join(d, "config.json")))
    text = {kk: j.pop(kk) for kk in ["hidden_size", "num_hidden_layers", "num_attention_heads", "num_key_value_heads", "intermediate_size",
                                    "vocab_size", "rms_norm_eps", "rope_theta", "rope_scaling", "tie_word_embeddings"]}
    text["rope_parameters"] = {"rope_type": "default", "rope_theta": text.pop("rope_theta"), "mrope_section": text.pop("rope_scaling")["mrope_section"]}
    j["text_config"] = text
    d2 = str(tmp_path / "ckpt2")
    os.makedirs(d2)
    json.dump(j, open(os.path.join(d2, "config.json"), "w"))
    cfg2, vc2, _ = read_checkpoint_configs(d2)
    cfg2.eos_token_ids = cfg.eos_token_ids
    assert cfg2 == cfg and vc2.depth == vc.depth and vc2.max_pixels == 28 * 28 * 1280      # no preprocessor_config: defaults
    # tokenizer: the vision special tokens are single ids, one <|image_pad|> per image
    tok = load_tokenizer(d)
    ids = tok("w20 w21 <|vision_start|><|image_pad|><|vision_end|> w22")["input_ids"]
    assert ids == [20, 21, 6, 5, 7, 22]
    assert tok.decode([20, 3, 21], skip_special_tokens=True) == "w20 w21"
    assert load_tokenizer(d2) is None
