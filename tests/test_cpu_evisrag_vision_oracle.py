"""The vision-tower oracle (oracle/qwen_vision_oracle.py) against the fixtures the HuggingFace Qwen2.5-VL
implementation produced in the build container (oracle/gen_golden_evisrag_vision.py ->
tests/golden/evisrag_vision_tiny.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle.qwen_gen_oracle import QwenGenOracle, synth_weights, tiny_config
from oracle.qwen_vision_oracle import (QwenVisionOracle, image_bounds, patchify, position_hw, smart_resize, tiny_vision_config,
                                       vision_weight_specs, window_order)
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
