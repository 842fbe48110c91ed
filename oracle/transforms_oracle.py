"""CPU ORACLE (test infrastructure, NOT product code) for the input side of the hot path:
what torchmultimodal/transforms/clip_transform.py:301-352 (CLIPImageTransform) and flava_transform.py:109-297
(FLAVAImageTransform's image branch) compute for one image.

The arithmetic of that transform lives in two third-party dependencies of the reference that are not vendored in /root/reference:
  * Pillow (unpinned by the reference; installed here: 12.2.0) -- `Image.resize(size, BICUBIC)` on an RGB image =
    src/libImaging/Resample.c: `precompute_coeffs` (double-precision bicubic weights, a = -0.5, support 2 * max(scale, 1), window
    [int(center - support + 0.5), int(center + support + 0.5)) clipped to the image, weights normalised by their running sum),
    `normalize_coeffs_8bpc` (fixed point, 22 fractional bits, rounded half away from zero), then a horizontal and a vertical
    pass per band, each accumulating from 1 << 21 in 32-bit integers and clipping (acc >> 22) to 0..255, with a uint8
    intermediate image between the passes.
  * torchvision (unpinned; NOT installed here) -- transforms.Resize(int) = shorter edge to `size`, longer edge
    int(size * long / short), no-op when the shorter edge already matches (functional._compute_resized_output_size);
    CenterCrop = box at int(round((H - h) / 2.0)), int(round((W - w) / 2.0)); ToTensor = uint8 HWC -> float32 CHW / 255;
    Normalize = (x - mean) / std in float32; RandomResizedCrop.get_params (10 tries of area * U(scale), exp(U(log ratio)), then the
    centred fallback) followed by crop + resize.

Pinning: `pil_resize` (bicubic and Lanczos) is checked bit for bit against Pillow itself (tests/test_oracle_transforms.py, random images over
up- and down-scales, extreme aspect ratios, 1-pixel edges); `tv_resize_output_size` against the reference's own KAT
(tests/transforms/test_clip_transform.py:141-149: 500x300 -> (373, 224)) and, as a second opinion, against Hugging Face
transformers' own restatement of torchvision's Resize (image_transforms.get_resize_output_image_size) over random sizes.  The torchvision parts have no executable reference in
this container: CenterCrop / ToTensor / Normalize are restated from their published definitions ("parity unpinned" for those three
one-liners and for the RandomResizedCrop parameter draw).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x: float) -> float:
    """Resample.c lanczos_filter: truncated sinc, a = 3."""
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


_FILTERS = {"bicubic": (_bicubic, 2.0), "lanczos": (_lanczos, 3.0)}


def pil_resample_coeffs(in_size: int, out_size: int, filter: str = "bicubic"):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full axis (box = whole image).
    Returns (kk int32 [out_size, ksize], bounds int32 [out_size, 2] = (first source index, tap count))."""
    kernel, filter_support = _FILTERS[filter]
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = filter_support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [kernel((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def _resample_rows(img: np.ndarray, out_size: int, filter: str = "bicubic") -> np.ndarray:
    """One 8-bit pass along axis 0 of a uint8 [N, M, C] array."""
    kk, bounds = pil_resample_coeffs(img.shape[0], out_size, filter)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    for i in range(out_size):
        x0, n = int(bounds[i, 0]), int(bounds[i, 1])
        acc = (img[x0:x0 + n].astype(np.int64) * kk[i, :n, None, None].astype(np.int64)).sum(0) + (1 << (PRECISION_BITS - 1))
        acc = ((acc + 2 ** 31) % 2 ** 32) - 2 ** 31  # the C accumulator is a 32-bit int
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


def pil_resize(img: np.ndarray, out_h: int, out_w: int, filter: str = "bicubic") -> np.ndarray:
    """Image.fromarray(img).resize((out_w, out_h), BICUBIC | LANCZOS) for uint8 [H, W, C]: horizontal pass, then vertical; a pass
    whose size does not change is skipped (ImagingResample)."""
    assert img.dtype == np.uint8 and img.ndim == 3
    t = img
    if out_w != img.shape[1]:
        t = _resample_rows(np.ascontiguousarray(t.transpose(1, 0, 2)), out_w, filter).transpose(1, 0, 2)
    if out_h != img.shape[0]:
        t = _resample_rows(np.ascontiguousarray(t), out_h, filter)
    return np.ascontiguousarray(t)


def pil_resize_bicubic(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    return pil_resize(img, out_h, out_w, "bicubic")


def tv_resize_output_size(h: int, w: int, size):
    """transforms.Resize(size) output (h, w): int = shorter edge to size; (h, w) tuple = exact."""
    if not isinstance(size, int):
        if len(size) == 1:
            size = int(size[0])
        else:
            return int(size[0]), int(size[1])
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def center_crop_box(h: int, w: int, ch: int, cw: int):
    """transforms.CenterCrop((ch, cw)) on an image at least that large: (top, left)."""
    return int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))


def to_tensor_normalize(img: np.ndarray, mean, std) -> np.ndarray:
    """ToTensor + Normalize: uint8 [H, W, 3] -> float32 [3, H, W]."""
    x = img.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return ((x - m) / s).astype(np.float32)


def clip_image_transform_eval(img: np.ndarray, image_size=224, mean=(0.48145466, 0.4578275, 0.40821073),
                              std=(0.26862954, 0.26130258, 0.27577711)) -> np.ndarray:
    """CLIPImageTransform(is_train=False) on one RGB uint8 [H, W, 3] image (clip_transform.py:339-345)."""
    oh, ow = tv_resize_output_size(img.shape[0], img.shape[1], image_size)
    r = pil_resize_bicubic(img, oh, ow)
    ch, cw = (image_size, image_size) if isinstance(image_size, int) else image_size
    top, left = center_crop_box(oh, ow, ch, cw)
    return to_tensor_normalize(r[top:top + ch, left:left + cw], mean, std)


def resized_crop(img: np.ndarray, i: int, j: int, h: int, w: int, image_size=224, mean=(0.48145466, 0.4578275, 0.40821073),
                 std=(0.26862954, 0.26130258, 0.27577711)) -> np.ndarray:
    """The training branch for a given crop box (clip_transform.py:332-337): crop, resize to image_size, ToTensor, Normalize."""
    oh, ow = (image_size, image_size) if isinstance(image_size, int) else image_size
    return to_tensor_normalize(pil_resize_bicubic(np.ascontiguousarray(img[i:i + h, j:j + w]), oh, ow), mean, std)


def patchify(x: np.ndarray, p: int) -> np.ndarray:
    """float [3, H, W] -> [(H/p)*(W/p), 3*p*p], column (c*p + py)*p + px: the im2col of the patch-embedding conv."""
    c, h, w = x.shape
    return x.reshape(c, h // p, p, w // p, p).transpose(1, 3, 0, 2, 4).reshape((h // p) * (w // p), c * p * p)


# ---------------------------------------------------------------------------------------------------------------- FLAVA
def map_pixels(x: np.ndarray, eps: float = 0.1) -> np.ndarray:
    """flava_transform.py:24-28 on a float32 array: the Python scalars enter torch's fp32 kernels as fp32, product then sum."""
    return (np.float32(1 - 2 * eps) * x.astype(np.float32)).astype(np.float32) + np.float32(eps)


def to_tensor(img: np.ndarray) -> np.ndarray:
    return img.astype(np.float32).transpose(2, 0, 1) / np.float32(255)


def flava_image_transform_eval(img: np.ndarray, encoder_size=224, codebook_size=112, mean=(0.48145466, 0.4578275, 0.40821073),
                               std=(0.26862954, 0.26130258, 0.27577711)):
    """FLAVAImageTransform(is_train=False) on one RGB uint8 image (flava_transform.py:109-150, 262-297): (image, image_for_codebook).
    TwoWayResize: exact (S, S) bicubic, then the RESIZED image to (C, C) with Lanczos."""
    first = pil_resize(img, encoder_size, encoder_size, "bicubic")
    second = pil_resize(first, codebook_size, codebook_size, "lanczos")
    return to_tensor_normalize(first, mean, std), map_pixels(to_tensor(second))


def flava_image_transform_train(img: np.ndarray, box, encoder_size=224, codebook_size=112, mean=(0.48145466, 0.4578275, 0.40821073),
                                std=(0.26862954, 0.26130258, 0.27577711)):
    """The training branch for a given crop box (i, j, h, w) (flava_transform.py:185-209): both sizes from the original crop."""
    i, j, h, w = box
    crop = np.ascontiguousarray(img[i:i + h, j:j + w])
    first = pil_resize(crop, encoder_size, encoder_size, "bicubic")
    second = pil_resize(crop, codebook_size, codebook_size, "lanczos")
    return to_tensor_normalize(first, mean, std), map_pixels(to_tensor(second))
