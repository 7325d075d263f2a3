"""GPU side of the MiniCPM-V slicing policy: Pillow-exact bicubic resize + slice cropping on the
device.  Same policy as preprocess.slice_image
(modeling_minicpmv.py:482-537); the resized pixels are bit-identical to PIL's, so embeddings do
not depend on which side resized."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from .preprocess import (PreparedItem, choose_grid, find_best_resize, get_grid_placeholder, get_refine_size,
                         image_placeholder)


def resize_bicubic(img, size: Tuple[int, int], device: int = 0) -> torch.Tensor:
    """img: uint8 HWC numpy array or cuda tensor; size = (out_w, out_h) like PIL.  -> cuda uint8 [oh, ow, 3]."""
    lib = _lib.load()
    ow, oh = int(size[0]), int(size[1])
    out = torch.empty((oh, ow, 3), dtype=torch.uint8, device=f"cuda:{device}")
    if isinstance(img, torch.Tensor):
        assert img.is_cuda and img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3
        img = img.contiguous()
        H, W = int(img.shape[0]), int(img.shape[1])
        src, on_dev = C.c_void_p(img.data_ptr()), 1
    else:
        a = np.ascontiguousarray(img, dtype=np.uint8)
        assert a.ndim == 3 and a.shape[2] == 3
        H, W = a.shape[:2]
        src, on_dev = C.c_void_p(a.ctypes.data), 0
    _lib.check(lib.vr_resize_bicubic(device, src, on_dev, H, W, C.c_void_p(out.data_ptr()), oh, ow,
                                     C.c_void_p(int(torch.cuda.current_stream(int(device)).cuda_stream))), "vr_resize_bicubic")
    return out


class PageUploader:
    """One pinned staging buffer + one device buffer per user (a pipeline slot): the u8 pixels of a whole batch of pages
    cross the bus in ONE asynchronous copy on the current stream; the pages come back as views of the device buffer.
    (Per-page uploads cost a pageable-memory copy and a stream synchronisation each — with batches in flight that wait is
    the previous batch's encode; `tensor.pin_memory()` per page is slower still: 1.6 ms per page measured.)"""

    def __init__(self, device: int = 0):
        self.device = int(device)
        self._pin: Optional[torch.Tensor] = None
        self._ev: Optional[torch.cuda.Event] = None

    def upload(self, images) -> List[Optional[torch.Tensor]]:
        arrs = [None if im is None else (im if isinstance(im, torch.Tensor) else
                                         np.asarray(im.convert("RGB") if hasattr(im, "convert") else im, dtype=np.uint8)) for im in images]
        host = [a for a in arrs if a is not None and not isinstance(a, torch.Tensor)]
        total = sum(int(a.size) for a in host)
        if total == 0:
            return arrs
        if self._ev is not None:
            self._ev.synchronize()                         # the previous upload has left the staging buffer (long ago)
        if self._pin is None or self._pin.numel() < total:
            self._pin = torch.empty(max(total, 1 << 24), dtype=torch.uint8, pin_memory=True)
        pin_np = self._pin.numpy()
        off, spans = 0, []
        for a in arrs:
            if a is None or isinstance(a, torch.Tensor):
                spans.append(None)
                continue
            n = int(a.size)
            pin_np[off:off + n] = a.reshape(-1)
            spans.append((off, n, a.shape))
            off += n
        dev = torch.empty(total, dtype=torch.uint8, device=f"cuda:{self.device}")
        dev.copy_(self._pin[:total], non_blocking=True)
        self._ev = torch.cuda.Event()
        self._ev.record(torch.cuda.current_stream(self.device))
        return [a if sp is None else dev[sp[0]:sp[0] + sp[1]].view(sp[2]) for a, sp in zip(arrs, spans)]


def to_device_u8(img, device: int = 0) -> torch.Tensor:
    """PIL image / u8 HWC array -> cuda uint8 tensor (a cuda tensor passes through)."""
    if isinstance(img, torch.Tensor):
        return img
    a = np.asarray(img.convert("RGB") if hasattr(img, "convert") else img, dtype=np.uint8)
    return torch.from_numpy(np.array(a, copy=True)).to(f"cuda:{device}")


def slice_image_gpu(img, cfg, device: int = 0):
    """-> (list of cuda uint8 HWC slices [source, patches row-major...], best_grid|None)."""
    img = to_device_u8(img, device)
    H, W = int(img.shape[0]), int(img.shape[1])
    size = (W, H)
    grid = choose_grid(size, cfg.max_slice_nums, cfg.scale_resolution)
    if grid is None:
        best = find_best_resize(size, cfg.scale_resolution, cfg.patch_size, allow_upscale=True)
        if tuple(best) == size:
            return [img.contiguous()], None               # already at its encode size: the upload is the slice
        return [resize_bicubic(img, best, device)], None
    src = resize_bicubic(img, find_best_resize(size, cfg.scale_resolution, cfg.patch_size), device)
    refine = get_refine_size(size, grid, cfg.scale_resolution, cfg.patch_size, allow_upscale=True)
    refined = resize_bicubic(img, refine, device)
    cw, ch = int(refine[0] / grid[0]), int(refine[1] / grid[1])
    out = [src]
    for y in range(0, refine[1], ch):
        for x in range(0, refine[0], cw):
            out.append(refined[y:y + ch, x:x + cw].contiguous())      # strided device copy (plumbing)
    return out, grid


def prepare_item_gpu(text: str, image, tokenizer, cfg, max_inp_length: Optional[int] = 2048, device: int = 0):
    """GPU-preprocessing twin of preprocess.prepare_item: returns (PreparedItem whose slices are
    uint8 cuda tensors, the same list of device slices)."""
    from .preprocess import prepare_item
    dev_slices: List[torch.Tensor] = []
    if image is None:
        return prepare_item(text, None, tokenizer, cfg, max_inp_length), dev_slices
    ph = image_placeholder(tokenizer, cfg.query_num)
    if cfg.slice_mode:
        dev_slices, grid = slice_image_gpu(image, cfg, device)
        if grid is not None:
            ph += get_grid_placeholder(tokenizer, grid, cfg.query_num)
    else:   # slice_mode=False: the image as it is, one plain placeholder (modeling_visrag_ret.py:70-72)
        dev_slices = [to_device_u8(image, device).contiguous()]
    it = prepare_item(ph + "\n" + text, None, tokenizer, cfg, max_inp_length)
    it.slices = list(dev_slices)        # device tensors: HipEncoder.encode_items passes them on as they are
    return it, dev_slices
