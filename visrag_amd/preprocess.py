"""Host-side pre-processing of the VisRAG-Ret encode path: MiniCPM-V slicing policy,
placeholder prompt, tokenisation, image bounds.  Stays on the CPU;
for BASELINE's 448x448 pages the resize is the identity and there is exactly one slice.

Mirrors (same names / argument meaning, own formulation):
  * slice_image / find_best_resize / get_refine_size / split_to_patches /
    get_grid_placeholder          modeling_minicpmv.py:482-609
  * VisRAG_Ret.prepare_context    modeling_visrag_ret.py:57-84
  * MiniCPMV._convert_to_tensors  modeling_minicpmv.py:173-200 (ids, truncation, image_bound)
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np


def ensure_divide(length: float, patch_size: int) -> int:
    return max(round(length / patch_size) * patch_size, patch_size)


def find_best_resize(original_size, scale_resolution: int, patch_size: int,
                     allow_upscale: bool = False) -> Tuple[int, int]:
    width, height = original_size
    if allow_upscale or width * height > scale_resolution * scale_resolution:
        aspect = width / height
        height = int(scale_resolution / math.sqrt(aspect))
        width = int(height * aspect)
    return ensure_divide(width, patch_size), ensure_divide(height, patch_size)


def get_refine_size(original_size, grid, scale_resolution: int, patch_size: int,
                    allow_upscale: bool = False) -> Tuple[int, int]:
    width, height = original_size
    gx, gy = grid
    cell = (ensure_divide(width, gx) / gx, ensure_divide(height, gy) / gy)
    bw, bh = find_best_resize(cell, scale_resolution, patch_size, allow_upscale=allow_upscale)
    return bw * gx, bh * gy


def choose_grid(size, max_slice_nums: int, scale_resolution: int) -> Optional[List[int]]:
    """None when the image is not split (<= one 448^2 area); else [cols, rows]."""
    w, h = size
    multiple = min(math.ceil(w * h / (scale_resolution * scale_resolution)), max_slice_nums)
    if multiple <= 1:
        return None
    log_ratio = math.log(w / h)
    best, best_err = [1, 1], float("inf")
    for n in (multiple - 1, multiple, multiple + 1):
        if n == 1 or n > max_slice_nums:
            continue
        for m in range(1, n + 1):
            if n % m:
                continue
            err = abs(log_ratio - math.log(m / (n // m)))
            if err < best_err:
                best, best_err = [m, n // m], err
    return best


def slice_image(image, max_slice_nums: int = 9, scale_resolution: int = 448, patch_size: int = 14):
    """PIL image -> (source_image, patches[rows][cols], best_grid).  Same contract as the
    reference's slice_image (modeling_minicpmv.py:482-537)."""
    from PIL import Image
    size = image.size
    grid = choose_grid(size, max_slice_nums, scale_resolution)
    if grid is None:
        best = find_best_resize(size, scale_resolution, patch_size, allow_upscale=True)
        src = image if tuple(best) == tuple(size) else image.resize(best, Image.Resampling.BICUBIC)
        return src, [], None
    src = image.resize(find_best_resize(size, scale_resolution, patch_size), Image.Resampling.BICUBIC)
    refine = get_refine_size(size, grid, scale_resolution, patch_size, allow_upscale=True)
    refined = image.resize(refine, Image.Resampling.BICUBIC)
    cw, ch = int(refine[0] / grid[0]), int(refine[1] / grid[1])
    patches = [[refined.crop((x, y, x + cw, y + ch)) for x in range(0, refine[0], cw)]
               for y in range(0, refine[1], ch)]
    return src, patches, grid


def image_placeholder(tokenizer, query_num: int) -> str:
    return tokenizer.im_start + tokenizer.unk_token * query_num + tokenizer.im_end


def get_grid_placeholder(tokenizer, grid, query_num: int) -> str:
    cols, rows = grid
    one = image_placeholder(tokenizer, query_num)
    return tokenizer.slice_start + "\n".join(one * cols for _ in range(rows)) + tokenizer.slice_end


@dataclass
class PreparedItem:
    input_ids: List[int]
    image_bound: List[Tuple[int, int]]
    slices: List[np.ndarray] = field(default_factory=list)   # u8 HWC, source first


def prepare_item(text: str, image, tokenizer, cfg, max_inp_length: Optional[int] = 2048) -> PreparedItem:
    """One (text, image|None) pair -> token ids, image bounds and u8 slices."""
    if not isinstance(text, str):
        raise NotImplementedError(f"chatml format expected, expect outmost type to be str but got {type(text)}")
    content, slices = text, []
    if image is not None and image:
        if cfg.slice_mode:
            src, patches, grid = slice_image(image, cfg.max_slice_nums, cfg.scale_resolution, cfg.patch_size)
            pil_slices = [src] + [p for row in patches for p in row]
            ph = image_placeholder(tokenizer, cfg.query_num)
            if patches:
                ph += get_grid_placeholder(tokenizer, grid, cfg.query_num)
        else:
            pil_slices = [image]
            ph = image_placeholder(tokenizer, cfg.query_num)
        content = ph + "\n" + content
        slices = [np.asarray(s if s.mode == "RGB" else s.convert("RGB"), dtype=np.uint8) for s in pil_slices]
    ids = tokenizer.encode(content)
    if not getattr(tokenizer, "add_bos_token", True):
        ids = [tokenizer.bos_id] + list(ids)
    ids = list(ids)
    if max_inp_length is not None:
        ids = ids[:max_inp_length]
    arr = np.asarray(ids, dtype=np.int64)
    starts = np.nonzero(arr == tokenizer.im_start_id)[0] + 1
    ends = np.nonzero(arr == tokenizer.im_end_id)[0]
    n = max(len(starts), len(ends))
    if len(starts) != len(ends):
        # the reference hstack()s the two lists and fails on a truncated placeholder;
        # keep only complete bounds but keep the error behaviour explicit
        raise ValueError("unbalanced <image> / </image> markers after truncation "
                         f"({len(starts)} starts, {len(ends)} ends); raise max_inp_length")
    bound = [(int(s), int(e)) for s, e in zip(starts[:n], ends[:n])]
    return PreparedItem(input_ids=ids, image_bound=bound, slices=slices)


_pool, _pool_workers = None, 0
_pool_lock = __import__("threading").Lock()


def prepare_batch(texts: Sequence[str], images: Sequence, tokenizer, cfg,
                  max_inp_length: Optional[int] = 2048, max_workers: int = 8) -> List[PreparedItem]:
    if len(texts) != len(images):
        raise ValueError("text and image lists must have the same length")
    # The reference prepares a batch on an 8-thread pool (modeling_visrag_ret.py:98).  That pays when pages are RESIZED or
    # sliced (Pillow's resampling releases the GIL); items that only tokenise and copy pixels — text, pages already at
    # scale_resolution — are GIL-bound, and the pool then costs more than it buys (32 pages of 448 x 448: 36 ms on eight
    # threads, 16 ms on one)
    def light(im):
        return im is None or not im or (getattr(im, "size", None) == (cfg.scale_resolution, cfg.scale_resolution))
    if len(texts) <= 1 or max_workers <= 1 or all(light(im) for im in images):
        return [prepare_item(t, im, tokenizer, cfg, max_inp_length) for t, im in zip(texts, images)]
    with _pool_lock:
        global _pool, _pool_workers
        if _pool is None or _pool_workers != max_workers:
            from concurrent.futures import ThreadPoolExecutor
            _pool, _pool_workers = ThreadPoolExecutor(max_workers=max_workers), max_workers
        pool = _pool
    return list(pool.map(lambda ti: prepare_item(ti[0], ti[1], tokenizer, cfg, max_inp_length), zip(texts, images)))
