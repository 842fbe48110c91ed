"""Host side of mmamd_image_resample: Pillow's resampling coefficient tables (src/libImaging/Resample.c precompute_coeffs +
normalize_coeffs_8bpc; bicubic a = -0.5, and Lanczos-3) for a window of output positions of one axis, built vectorised over the outputs in the
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


def _lanczos(x: np.ndarray) -> np.ndarray:
    """Resample.c lanczos_filter = sinc(x) * sinc(x / 3) on [-3, 3), with libm's sin through math.sin (Pillow links the same libm;
    numpy's vectorised sin may differ in the last bit)."""
    def sinc(v: float) -> float:
        if v == 0.0:
            return 1.0
        v = v * math.pi
        return math.sin(v) / v

    out = np.zeros(x.shape, np.float64)
    flat, o = x.reshape(-1), out.reshape(-1)
    for i in range(flat.size):
        v = float(flat[i])
        if -3.0 <= v < 3.0:
            o[i] = sinc(v) * sinc(v / 3)
    return out


FILTERS = {"bicubic": (_bicubic, 2.0), "lanczos": (_lanczos, 3.0)}


@lru_cache(maxsize=4096)
def axis_tables(in_size: int, out_size: int, first: int, count: int, filter: str = "bicubic") -> Tuple[np.ndarray, np.ndarray]:
    """Fixed-point coefficients int32 [count, ksize] and (first source index, taps) int32 [count, 2] for the output positions
    [first, first + count) of an axis resized in_size -> out_size.  The returned arrays are cached: do not write to them."""
    kernel, filter_support = FILTERS[filter]
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = filter_support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = 0.0 + (np.arange(first, first + count, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    taps = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.int64)[None, :]
    arg = ((x + xmin[:, None]).astype(np.float64) - center[:, None] + 0.5) * ss
    w = kernel(np.where(x < taps[:, None], arg, 1e9))  # 1e9: outside every filter's support -> weight 0, never evaluated
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


def normalize_lut(mean, std) -> np.ndarray:
    """float32 [3, 256]: torchvision ToTensor + Normalize of byte v in channel c, ((v / 255) - mean[c]) / std[c], every operation
    rounded to fp32 as torch's CPU kernels do (uint8 -> float32, div by 255, sub_, div_)."""
    x = np.arange(256, dtype=np.float32) / np.float32(255)
    m = np.asarray(mean, np.float32)[:, None]
    s = np.asarray(std, np.float32)[:, None]
    return ((x[None, :] - m) / s).astype(np.float32)


def map_pixels_lut(eps: float = 0.1) -> np.ndarray:
    """float32 [3, 256]: ToTensor then flava_transform.map_pixels, (1 - 2 eps) * (v / 255) + eps with the Python scalars cast to
    fp32 and the product rounded before the sum (flava_transform.py:24-28, 280-285)."""
    x = np.arange(256, dtype=np.float32) / np.float32(255)
    y = (np.float32(1 - 2 * eps) * x).astype(np.float32) + np.float32(eps)
    return np.broadcast_to(y.astype(np.float32), (3, 256)).copy()
