"""-m gpu: Pillow-exact bicubic resize + slicing on the device.
Integer / byte work: the bar is BIT-EXACT against Pillow."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from visrag_amd.config import tiny_config, full_config  # noqa: E402
from visrag_amd.gpu_resize import prepare_item_gpu, resize_bicubic, slice_image_gpu  # noqa: E402
from visrag_amd.preprocess import prepare_item, slice_image  # noqa: E402
from visrag_amd.tokenizer import StandInTokenizer  # noqa: E402


@pytest.mark.parametrize("hw,size", [((300, 200), (140, 84)), ((200, 300), (364, 546)), ((448, 448), (448, 448)),
                                     ((64, 50), (518, 392)), ((333, 517), (112, 112)), ((100, 100), (100, 37)),
                                     ((97, 131), (210, 131)), ((2339, 1654), (378, 532)), ((1670, 1114), (1036, 1568))])
def test_resize_bit_exact_vs_pillow(hw, size):
    from PIL import Image
    rng = np.random.default_rng(hw[0] * 7 + size[0])
    img = rng.integers(0, 256, size=(hw[0], hw[1], 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize(size, Image.Resampling.BICUBIC))
    got_host = resize_bicubic(img, size).cpu().numpy()
    got_dev = resize_bicubic(torch.from_numpy(img).cuda(), size).cpu().numpy()
    assert np.array_equal(got_host, ref)
    assert np.array_equal(got_dev, ref)


@pytest.mark.parametrize("wh", [(448, 448), (1114, 1670), (1654, 2339), (1072, 670), (564, 3040), (200, 150)])
def test_slice_image_gpu_matches_host_policy(wh):
    from PIL import Image
    cfg = full_config()
    rng = np.random.default_rng(wh[0])
    img = Image.fromarray(rng.integers(0, 256, size=(wh[1], wh[0], 3), dtype=np.uint8))
    src, patches, grid = slice_image(img, cfg.max_slice_nums, cfg.scale_resolution, cfg.patch_size)
    host = [np.asarray(src)] + [np.asarray(p) for row in patches for p in row]
    dev, g2 = slice_image_gpu(img, cfg)
    assert grid == g2 and len(dev) == len(host)
    for a, b in zip(host, dev):
        assert np.array_equal(a, b.cpu().numpy())


def test_prepare_item_gpu_equals_host():
    from PIL import Image
    cfg = tiny_config()
    tok = StandInTokenizer(cfg.vocab_size)
    rng = np.random.default_rng(1)
    img = Image.fromarray(rng.integers(0, 256, size=(200, 300, 3), dtype=np.uint8))
    h = prepare_item("caption", img, tok, cfg, 2048)
    g, dev = prepare_item_gpu("caption", img, tok, cfg, 2048)
    assert g.input_ids == h.input_ids and g.image_bound == h.image_bound and len(dev) == len(h.slices)
    for a, b in zip(h.slices, dev):
        assert np.array_equal(a, b.cpu().numpy())


def test_model_embeddings_do_not_depend_on_where_the_page_was_resized():
    """DRModelForInference(gpu_preprocess=True) == (gpu_preprocess=False): same slices bit for
    bit, hence the same embeddings bit for bit (sliced and unsliced pages in one batch)."""
    from PIL import Image
    from visrag_amd.modeling import DRModelForInference
    from visrag_amd.synth import iter_synth_weights
    cfg = tiny_config()
    tok = StandInTokenizer(cfg.vocab_size)
    m = DRModelForInference.build(cfg=cfg, state_dict=iter_synth_weights(cfg, 0, device="cuda"), max_images=16)
    rng = np.random.default_rng(2)
    imgs = [Image.fromarray(rng.integers(0, 256, size=hw + (3,), dtype=np.uint8)) for hw in [(112, 112), (200, 300), (90, 400)]]
    batch = {"text": ["", "a caption", ""], "image": imgs}
    m.gpu_preprocess = True
    a = m(passage=batch, tokenizer=tok).p_reps.cpu().numpy()
    m.gpu_preprocess = False
    b = m(passage=batch, tokenizer=tok).p_reps.cpu().numpy()
    assert np.array_equal(a, b)
