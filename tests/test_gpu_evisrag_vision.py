"""EVisRAG vision tower on the GPU against the HF-generated fixture (oracle/gen_golden_evisrag_vision.py ->
tests/golden/evisrag_vision_tiny.npz) and the oracle: embedding rows of three pages of different shapes (full, ragged
and single attention windows), the whole prompt-with-images path down to the logits, the generate call site with
processor-style inputs."""
import os

import numpy as np
import pytest
import torch

from oracle.qwen_gen_oracle import QwenGenOracle, synth_weights, tiny_config
from oracle.qwen_vision_oracle import QwenVisionOracle, hd80_vision_config, tiny_vision_config

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "evisrag_vision_tiny.npz")


def _vision_weights(g):
    return {k[2:]: torch.from_numpy(g[k]).view(torch.bfloat16) for k in g.files if k.startswith("w:")}


def _pixels(g):
    return torch.from_numpy(g["pixels_bf16"]).view(torch.bfloat16).float().numpy()


@pytest.fixture(scope="module")
def setup():
    from visrag_amd.evisrag import LLM, GenConfig, VisionConfig
    g = np.load(GOLD)
    cfg, vcfg = tiny_config(), tiny_vision_config(256)
    gc = GenConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                   num_key_value_heads=cfg.num_key_value_heads, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
                   rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, mrope_section=tuple(cfg.mrope_section),
                   image_token_id=int(g["image_token_id"]), eos_token_ids=())
    vc = VisionConfig(depth=vcfg.depth, hidden_size=vcfg.hidden_size, num_heads=vcfg.num_heads, intermediate_size=vcfg.intermediate_size,
                      out_hidden_size=vcfg.out_hidden_size, window_size=vcfg.window_size,
                      fullatt_block_indexes=tuple(vcfg.fullatt_block_indexes), min_pixels=56 * 56, max_pixels=28 * 28 * 24)
    w = dict(synth_weights(cfg, seed=int(g["lm_seed"])))
    w.update(_vision_weights(g))                                      # "model.visual.*", bf16
    llm = LLM(gc, max_model_len=512, max_prefill=256, vision=vc, max_vision_rows=512, weights=w)
    yield g, cfg, vcfg, llm
    llm.close()


def _close(ours, ref, rel, cos_min):
    scale = np.abs(ref).max()
    assert np.abs(ours - ref).max() < rel * scale, (np.abs(ours - ref).max(), scale)
    num = (ours * ref).sum(-1)
    cos = num / (np.linalg.norm(ours, axis=-1) * np.linalg.norm(ref, axis=-1))
    assert cos.min() > cos_min, cos.min()


def test_tower_embeddings_match_hf(setup):
    g, cfg, vcfg, llm = setup
    emb = llm.encode_images(_pixels(g), g["grids"])
    assert emb.shape == g["image_embeds"].shape
    # bf16 GEMM operands and attention probabilities against HF's fp32 run; fp32 residual stream and accumulation
    _close(emb, g["image_embeds"], 2e-2, 1 - 1e-3)


def test_tower_with_head_dim_80_matches_hf():
    """The 7B tower's head_dim runs on the attention kernel's own head_dim-80 form (no 128-wide slots): a second HF
    fixture, four pages incl. one whose full-attention block spans several key tiles."""
    from visrag_amd.evisrag import LLM, GenConfig, VisionConfig
    g = np.load(os.path.join(os.path.dirname(GOLD), "evisrag_vision_hd80.npz"))
    cfg, vcfg = tiny_config(), hd80_vision_config(256)
    gc = GenConfig(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                   num_key_value_heads=cfg.num_key_value_heads, intermediate_size=cfg.intermediate_size, vocab_size=cfg.vocab_size,
                   rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, mrope_section=tuple(cfg.mrope_section), image_token_id=5,
                   eos_token_ids=())
    vc = VisionConfig(depth=vcfg.depth, hidden_size=vcfg.hidden_size, num_heads=vcfg.num_heads, intermediate_size=vcfg.intermediate_size,
                      out_hidden_size=vcfg.out_hidden_size, window_size=vcfg.window_size,
                      fullatt_block_indexes=tuple(vcfg.fullatt_block_indexes))
    w = dict(synth_weights(cfg, seed=7))
    w.update(_vision_weights(g))
    llm = LLM(gc, max_model_len=512, max_prefill=256, vision=vc, max_vision_rows=512, weights=w)
    try:
        emb = llm.encode_images(_pixels(g), g["grids"])
        _close(emb, g["image_embeds"], 2e-2, 1 - 1e-3)
    finally:
        llm.close()


def test_tower_per_image_equals_batched(setup):
    """Images do not see each other: every page alone gives the rows it gets in the three-page call."""
    g, cfg, vcfg, llm = setup
    px, grids = _pixels(g), g["grids"]
    allrows = llm.encode_images(px, grids)
    r0 = t0 = 0
    for gr in grids:
        n = int(gr[0] * gr[1] * gr[2])
        one = llm.encode_images(px[r0:r0 + n], gr[None])
        # (bit-equal as long as both calls pick the same GEMM tiles; a different tile may round the bf16 steps differently)
        np.testing.assert_allclose(one, allrows[t0:t0 + n // 4], rtol=0, atol=5e-3 * float(np.abs(allrows).max()))
        r0 += n
        t0 += n // 4


def test_prompt_with_images_logits_match_hf(setup):
    """ids with expanded placeholders + pixel rows -> tower -> rows dropped into the prompt on the device -> logits."""
    g, cfg, vcfg, llm = setup
    pos3, ids = llm.prefill_images(g["prompt_ids"].tolist(), _pixels(g), g["grids"])
    np.testing.assert_array_equal(pos3, g["prompt_pos3"])
    assert ids == g["prompt_ids"].tolist()
    _close(llm.logits()[None], g["prompt_logits"][None], 2e-2, 1 - 3e-4)


def test_generate_with_one_placeholder_per_image(setup):
    """predict.py:147's call shape: the prompt holds ONE placeholder per image (the chat template's <|image_pad|>),
    multi_modal_data carries the processor's output; the first sampled token is the argmax of the fixture's logits
    unless that is a near-tie."""
    from visrag_amd.evisrag import SamplingParams
    g, cfg, vcfg, llm = setup
    tid = int(g["image_token_id"])
    short, prev = [], None
    for t in g["prompt_ids"].tolist():                               # collapse every placeholder run to one token
        if t != tid or prev != tid:
            short.append(t)
        prev = t
    assert short.count(tid) == 3
    out = llm.generate([{"prompt_token_ids": short, "multi_modal_data": {"pixel_values": _pixels(g), "image_grid_thw": g["grids"]}}],
                       SamplingParams(temperature=0.0, repetition_penalty=1.0, max_tokens=3, stop_token_ids=()))
    toks = out[0].outputs[0].token_ids
    assert len(toks) == 3
    ref = g["prompt_logits"]
    top = np.argsort(ref)[::-1]
    assert toks[0] == top[0] or ref[top[0]] - ref[toks[0]] < 2e-2 * np.abs(ref).max()


def test_generate_from_pil_pages(setup):
    """PIL pages through process_images (smart_resize, bicubic, normalise, patchify), tower and language model against
    the two oracles chained the same way."""
    from PIL import Image
    from visrag_amd.evisrag import SamplingParams, process_images, rope_index
    g, cfg, vcfg, llm = setup
    rng = np.random.default_rng(5)
    pages = [Image.fromarray(rng.integers(0, 256, (150, 97, 3), dtype=np.uint8)), Image.fromarray(rng.integers(0, 256, (60, 120, 3), dtype=np.uint8))]
    tid = int(g["image_token_id"])
    ids = [20, 21, 6, tid, 7, 22, 6, tid, 7, 23, 24]
    px, grid = process_images(pages, llm.vision)
    pos3, full_ids = llm.prefill_images(ids, px, grid)
    ours = llm.logits()
    vo = QwenVisionOracle(vcfg, {k: v.float() for k, v in _vision_weights(g).items()})
    lm = QwenGenOracle(cfg, synth_weights(cfg, seed=int(g["lm_seed"])))
    grids = [tuple(int(v) for v in r) for r in grid]
    idt = torch.tensor(full_ids)
    emb = lm.embed(idt).clone()
    emb[idt == tid] = vo.forward(torch.from_numpy(px).to(torch.bfloat16).float(), grids)
    np.testing.assert_array_equal(pos3, rope_index(full_ids, tid, [(h // 2, w // 2) for _, h, w in grids]))
    ref = lm.forward(emb, torch.from_numpy(pos3).long())[-1].numpy()
    _close(ours[None], ref[None], 2e-2, 1 - 3e-4)
    out = llm.generate([{"prompt_token_ids": ids, "multi_modal_data": {"image": pages}}],
                       SamplingParams(temperature=0.0, repetition_penalty=1.05, max_tokens=2, stop_token_ids=()))
    assert len(out[0].outputs[0].token_ids) == 2 and out[0].prompt_token_ids == full_ids


def test_errors(setup):
    from visrag_amd._lib import VisragHipError
    g, cfg, vcfg, llm = setup
    with pytest.raises(VisragHipError):                               # odd grid: not a multiple of the merge size
        llm.encode_images(np.zeros((15, 1176), np.float32), np.array([[1, 3, 5]], np.int32))
    with pytest.raises(VisragHipError):                               # more rows than max_vision_rows
        llm.encode_images(np.zeros((24 * 24, 1176), np.float32), np.array([[1, 24, 24]], np.int32))
    with pytest.raises(ValueError):
        llm.prefill_images([1, 2, 3], _pixels(g), g["grids"])        # no placeholders for three images


def test_llm_from_checkpoint_directory_with_text_prompt(setup, tmp_path):
    """predict.py:112-147 as written: LLM(model=<directory>), a chat-templated prompt STRING with one <|image_pad|> per
    page, PIL pages in multi_modal_data, .outputs[0].text out — against the same model built from a GenConfig + weights
    and fed token ids."""
    from PIL import Image
    from tests.evisrag_ckpt_util import make_tiny_checkpoint
    from visrag_amd.evisrag import LLM, SamplingParams
    g, cfg, vcfg, llm = setup
    d = str(tmp_path / "ckpt")
    make_tiny_checkpoint(d)
    rng = np.random.default_rng(9)
    pages = [Image.fromarray(rng.integers(0, 256, (150, 97, 3), dtype=np.uint8)), Image.fromarray(rng.integers(0, 256, (60, 120, 3), dtype=np.uint8))]
    prompt = "w20 w21 <|vision_start|><|image_pad|><|vision_end|> w22 <|vision_start|><|image_pad|><|vision_end|> w23 w24"
    ids = [20, 21, 6, 5, 7, 22, 6, 5, 7, 23, 24]
    sp = SamplingParams(temperature=0.0, repetition_penalty=1.05, max_tokens=12)
    ck = LLM(model=d, tensor_parallel_size=1, dtype="bfloat16", limit_mm_per_prompt={"image": 5, "video": 0}, max_model_len=512,
             max_prefill=256, max_vision_rows=512)
    try:
        assert ck.cfg.eos_token_ids == (3, 4) and ck.vision.max_pixels == 28 * 28 * 24
        out = ck.generate([{"prompt": prompt, "multi_modal_data": {"image": pages}}], sampling_params=sp)[0]
    finally:
        ck.close()
    sp_ref = SamplingParams(temperature=0.0, repetition_penalty=1.05, max_tokens=12, stop_token_ids=(3, 4))
    ref = llm.generate([{"prompt_token_ids": ids, "multi_modal_data": {"image": pages}}], sp_ref)[0]
    assert out.prompt_token_ids == ref.prompt_token_ids
    assert out.outputs[0].token_ids == ref.outputs[0].token_ids
    assert out.outputs[0].text == " ".join(f"w{t}" for t in out.outputs[0].token_ids if t >= 8)


def test_pages_processed_on_the_gpu_equal_the_host_processor(setup):
    """vg_vision_encode_pages (Pillow-exact GPU resize + rescale / normalise / patchify inside the tower call) against the
    host image processor restatement (process_images: PIL resize + numpy) feeding vg_vision_encode: same bf16 patch
    rows, hence the same embedding rows; and generate() with PIL pages takes the GPU path by default."""
    from PIL import Image
    from visrag_amd.evisrag import process_images, process_pages_gpu
    g, cfg, vcfg, llm = setup
    rng = np.random.default_rng(7)
    pages = [Image.fromarray(rng.integers(0, 256, size=s + (3,), dtype=np.uint8)) for s in ((60, 90), (200, 130), (56, 56), (84, 112))]
    px, grid = process_images(pages, llm.vision)
    host = llm.encode_images(px, grid)
    dev_pages, grid2 = process_pages_gpu(pages, llm.vision, llm.device)
    assert np.array_equal(grid, grid2)
    assert all(t.is_cuda and t.dtype == torch.uint8 for t in dev_pages)
    dev = llm.encode_pages(dev_pages, grid2)
    assert dev.shape == host.shape
    scale = np.abs(host).max()
    assert np.abs(dev - host).max() < 2e-3 * scale, (np.abs(dev - host).max(), scale)
    assert (np.abs(dev - host).max(axis=1) == 0).mean() > 0.9            # bit-identical rows but for stray 1-ulp divisions
    # a page already at its processed size, handed over as a cuda tensor, is used in place
    t = torch.from_numpy(np.asarray(pages[2])).cuda()
    again, _ = process_pages_gpu([t], llm.vision, llm.device)
    assert again[0].data_ptr() == t.data_ptr()
    with pytest.raises(ValueError):
        llm.encode_pages([dev_pages[0]], grid2[1:2])
