"""TEST INFRASTRUCTURE ONLY — numpy restatement of Pillow's 8-bit bicubic resize
(src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
ImagingResampleHorizontal_8bpc / Vertical_8bpc; third-party, restated from the published source).
Pinned against Pillow itself in tests/test_cpu_host.py; it documents the arithmetic the HIP
kernels in visrag_amd/csrc/resize.hip implement."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def coeffs(in_size: int, out_size: int):
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            v = k * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if v < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """resample `img` (u8, HWC) along axis (1 = horizontal, 0 = vertical)."""
    in_size = img.shape[axis]
    bounds, kk = coeffs(in_size, out_size)
    src = np.moveaxis(img.astype(np.int64), axis, 0)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.int64)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = np.tensordot(kk[xx, :n], src[xmin:xmin + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[xx] = acc >> PRECISION_BITS
    return np.moveaxis(np.clip(out, 0, 255).astype(np.uint8), 0, axis)


def resize_bicubic(img: np.ndarray, size) -> np.ndarray:
    """img u8 [H,W,3]; size = (out_w, out_h).  Horizontal pass first, 8-bit intermediate."""
    ow, oh = size
    out = img
    if ow != img.shape[1]:
        out = _pass(out, ow, 1)
    if oh != img.shape[0]:
        out = _pass(out, oh, 0)
    return out.copy() if out is img else out
