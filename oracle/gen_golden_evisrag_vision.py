"""Generate tests/golden/evisrag_vision_tiny.npz from the HuggingFace Qwen2.5-VL implementation installed in the BUILD
container (run here, never on the GPU box):

    python oracle/gen_golden_evisrag_vision.py

A tiny Qwen2_5_VLForConditionalGeneration: the language model of oracle/qwen_gen_oracle.tiny_config with its synthetic
weights, and the vision tower of oracle/qwen_vision_oracle.tiny_vision_config with weights drawn here (stored in the
fixture).  Three page "images" of different shapes, so that the window order has full, ragged and single windows:
  10 x 6 patches (5 x 3 merged tokens: windows of 2 x 2 with ragged right and bottom edges),
   4 x 8 patches (2 x 4 merged tokens: exactly two windows — HF's padding adds empty ones),
   2 x 2 patches (one merged token).
Stored:
  * the HF image processor's output for a PIL image (pixel rows + grid) next to the normalised image it came from —
    pins smart_resize and patchify;
  * the tower's merged embedding rows for the three images in one call, and its per-row hidden state before the merger;
  * the whole model's last-token logits for a prompt "text, image 0, text, image 1, image 2, text" with HF's own
    get_rope_index positions — pins the placeholder replacement and the positions visrag_amd.evisrag.rope_index gives.
A second file, tests/golden/evisrag_vision_hd80.npz: a tower with the 7B model's head_dim 80 (hd80_vision_config), four
pages, embedding rows only.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.qwen_gen_oracle import synth_weights, tiny_config  # noqa: E402
from oracle.qwen_vision_oracle import hd80_vision_config, synth_vision_weights, tiny_vision_config  # noqa: E402

IMAGE_TOKEN, VISION_START, VISION_END = 5, 6, 7


def build_hf(cfg, vcfg, weights):
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    text = dict(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                num_key_value_heads=cfg.num_key_value_heads, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
                max_position_embeddings=cfg.max_position_embeddings, rms_norm_eps=cfg.rms_norm_eps, tie_word_embeddings=False,
                bos_token_id=None, eos_token_id=None,
                rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta, "mrope_section": list(cfg.mrope_section)})
    vision = dict(depth=vcfg.depth, hidden_size=vcfg.hidden_size, intermediate_size=vcfg.intermediate_size, num_heads=vcfg.num_heads,
                  out_hidden_size=vcfg.out_hidden_size, patch_size=vcfg.patch_size, spatial_merge_size=vcfg.spatial_merge_size,
                  temporal_patch_size=vcfg.temporal_patch_size, window_size=vcfg.window_size,
                  fullatt_block_indexes=list(vcfg.fullatt_block_indexes))
    m = Qwen2_5_VLForConditionalGeneration(Qwen2_5_VLConfig(
        text_config=text, vision_config=vision, bos_token_id=None, eos_token_id=None, image_token_id=IMAGE_TOKEN,
        video_token_id=8, vision_start_token_id=VISION_START, vision_end_token_id=VISION_END)).eval().float()
    sd = m.state_dict()
    for k, v in weights.items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
        sd[k].copy_(v)
    return m


def main():
    cfg = tiny_config()
    vcfg = tiny_vision_config(cfg.hidden_size)
    w = dict(synth_weights(cfg, seed=7))
    vw = synth_vision_weights(vcfg, seed=23)
    w.update(vw)
    m = build_hf(cfg, vcfg, w)
    g = torch.Generator().manual_seed(31)
    out = {}

    # 1. image processor: a PIL page -> pixel rows
    from PIL import Image
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil
    rgb = torch.randint(0, 256, (150, 97, 3), generator=g, dtype=torch.uint8).numpy()
    proc = Qwen2VLImageProcessorPil(min_pixels=56 * 56, max_pixels=28 * 28 * 24)
    pr = proc(images=[Image.fromarray(rgb)], return_tensors="np")
    out["proc_rgb"] = rgb
    out["proc_pixel_values"] = pr["pixel_values"].astype(np.float32)
    out["proc_grid"] = pr["image_grid_thw"].astype(np.int32)
    out["proc_min_max_pixels"] = np.array([56 * 56, 28 * 28 * 24])
    out["proc_mean"] = np.array(proc.image_mean, dtype=np.float32)
    out["proc_std"] = np.array(proc.image_std, dtype=np.float32)

    # 2. the tower
    grids = [(1, 10, 6), (1, 4, 8), (1, 2, 2)]
    rows = sum(t * h * ww for t, h, ww in grids)
    pixels = (torch.randn((rows, vcfg.patch_dim), generator=g)).to(torch.bfloat16).float()
    gt = torch.tensor(grids)
    with torch.no_grad():
        vo = m.model.visual(pixels, grid_thw=gt)
    out["grids"] = np.array(grids, dtype=np.int32)
    out["pixels_bf16"] = pixels.to(torch.bfloat16).view(torch.int16).numpy()      # bf16 bit patterns
    out["image_embeds"] = vo.pooler_output.numpy().astype(np.float32)
    out["tower_rows_window_order"] = vo.last_hidden_state.numpy().astype(np.float32)

    # 3. the whole model on a prompt with the three images
    mm = vcfg.spatial_merge_size
    n_tok = [t * (h // mm) * (ww // mm) for t, h, ww in grids]
    txt = lambda n: torch.randint(16, cfg.vocab_size, (n,), generator=g)       # noqa: E731
    img = lambda i: torch.cat([torch.tensor([VISION_START]), torch.full((n_tok[i],), IMAGE_TOKEN), torch.tensor([VISION_END])])  # noqa: E731
    ids = torch.cat([txt(7), img(0), txt(3), img(1), img(2), txt(9)])
    mm_type = (ids == IMAGE_TOKEN).int()
    with torch.no_grad():
        pos, _ = m.model.get_rope_index(ids[None], mm_token_type_ids=mm_type[None], image_grid_thw=gt)
        res = m(input_ids=ids[None], pixel_values=pixels, image_grid_thw=gt, mm_token_type_ids=mm_type[None])
    out["prompt_ids"] = ids.numpy().astype(np.int32)
    out["prompt_pos3"] = pos[:, 0].numpy().astype(np.int32)
    out["prompt_logits"] = res.logits[0, -1].numpy().astype(np.float32)
    out["image_token_id"] = np.array(IMAGE_TOKEN)
    for k, v in vw.items():
        out["w:" + k] = v.to(torch.bfloat16).view(torch.int16).numpy()          # bf16 bit patterns
    out["lm_seed"] = np.array(7)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "evisrag_vision_tiny.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), {k: v.shape for k, v in out.items() if not k.startswith("w:")})

    # 4. a tower with the 7B model's head_dim 80 (its own attention-kernel configuration in the product): tower only
    v80 = hd80_vision_config(cfg.hidden_size)
    w80 = synth_vision_weights(v80, seed=29)
    wl = dict(synth_weights(cfg, seed=7))
    wl.update(w80)
    m80 = build_hf(cfg, v80, wl)
    grids80 = [(1, 10, 6), (1, 4, 8), (1, 2, 2), (1, 18, 12)]       # the last one: 216 rows, several tiles of full attention
    rows80 = sum(t * h * ww for t, h, ww in grids80)
    px80 = torch.randn((rows80, v80.patch_dim), generator=g).to(torch.bfloat16)
    with torch.no_grad():
        vo80 = m80.model.visual(px80.float(), grid_thw=torch.tensor(grids80))
    out80 = {"grids": np.array(grids80, dtype=np.int32), "pixels_bf16": px80.view(torch.int16).numpy(),
             "image_embeds": vo80.pooler_output.numpy().astype(np.float32)}
    for k, v in w80.items():
        out80["w:" + k] = v.to(torch.bfloat16).view(torch.int16).numpy()
    dst80 = os.path.join(os.path.dirname(dst), "evisrag_vision_hd80.npz")
    np.savez_compressed(dst80, **out80)
    print("wrote", dst80, os.path.getsize(dst80))


if __name__ == "__main__":
    main()
