"""Host side of mmamd_image_resample: Pillow's resampling coefficient tables (src/libImaging/Resample.c precompute_coeffs +
normalize_coeffs_8bpc, bicubic a = -0.5) for a window of output positions of one axis, built vectorised over the outputs in the
same double-precision operation order as the C code, and the torchvision size / crop rules CLIPImageTransform composes
(clip_transform.py:332-345).  tests/test_oracle_transforms.py checks the tables against the loop restatement in
oracle/transforms_oracle.py, which itself is pinned to Pillow bit for bit."""
from __future__ import annotations

import math
from functools import lru_cache
from typing import Tuple, Union

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: np.ndarray) -> np.ndarray:
    x = np.abs(x)
    near = ((1.5 * x) - 2.5) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * -0.5
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


@lru_cache(maxsize=4096)
def axis_tables(in_size: int, out_size: int, first: int, count: int) -> Tuple[np.ndarray, np.ndarray]:
    """Fixed-point coefficients int32 [count, ksize] and (first source index, taps) int32 [count, 2] for the output positions
    [first, first + count) of an axis resized in_size -> out_size.  The returned arrays are cached: do not write to them."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = 0.0 + (np.arange(first, first + count, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    taps = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.int64)[None, :]
    w = _bicubic(((x + xmin[:, None]).astype(np.float64) - center[:, None] + 0.5) * ss)
    w = np.where(x < taps[:, None], w, 0.0)
    ww = np.zeros(count, np.float64)
    for k in range(ksize):  # the C loop's summation order; the masked taps add +0.0
        ww = ww + w[:, k]
    ok = ww != 0.0
    w = np.where(ok[:, None], w / np.where(ok, ww, 1.0)[:, None], w)
    q = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS))
    kk = np.trunc(q).astype(np.int32)
    bounds = np.stack([xmin, taps], axis=1).astype(np.int32)
    kk.setflags(write=False)
    bounds.setflags(write=False)
    return kk, bounds


def resize_output_size(h: int, w: int, size: Union[int, Tuple[int, ...]]) -> Tuple[int, int]:
    """torchvision transforms.Resize(size): an int sends the shorter edge to `size` and the longer one to int(size * long / short)
    (unchanged if the shorter edge already matches); a pair is the exact (h, w)."""
    if not isinstance(size, int):
        if len(size) == 1:
            size = int(size[0])
        else:
            return int(size[0]), int(size[1])
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_long = int(size * long / short)
    return (new_long, size) if w <= h else (size, new_long)


def center_crop_origin(h: int, w: int, ch: int, cw: int) -> Tuple[int, int]:
    """torchvision transforms.CenterCrop: (top, left) of the ch x cw box (Python round: half to even)."""
    return int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))
