"""The device-side resampling engine behind CLIPImageTransform and FLAVAImageTransform: packs a ragged batch of decoded uint8
images, their geometry descriptors, the Pillow coefficient tables and the byte -> float value table into ONE pinned staging
buffer, copies it once, and launches mmamd_image_resample (multimodal_amd/csrc/image.hip).  Replaces the per-image PIL + torch host
loop of torchmultimodal/transforms/clip_transform.py:349-352 / flava_transform.py:286-311.  No CPU path: without a HIP device it
raises."""
from __future__ import annotations

import math
from functools import lru_cache
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from torch import Tensor

from .. import ops
from ._resample import axis_tables

_DESC = 16  # int64 words per image (include/mmamd.h, mmamd_image_resample)
_COPY_THREADS = 4
_pool = None


def _stage_pixels(sn: np.ndarray, jobs) -> None:
    """Copy host images into the pinned staging array: jobs = [(byte offset, uint8 array)].  numpy releases the GIL inside a large
    copy, so a few threads split a big batch (the copy is the serial part of the host path: ~0.56 MB per 500x375 image)."""
    global _pool
    total = sum(a.size for _, a in jobs)
    if total < (8 << 20) or len(jobs) < 2 * _COPY_THREADS:
        for o, a in jobs:
            sn[o:o + a.size] = a.reshape(-1)
        return
    if _pool is None:
        from concurrent.futures import ThreadPoolExecutor

        _pool = ThreadPoolExecutor(max_workers=_COPY_THREADS, thread_name_prefix="mmamd-stage")

    def part(chunk):
        for o, a in chunk:
            sn[o:o + a.size] = a.reshape(-1)

    step = (len(jobs) + _COPY_THREADS - 1) // _COPY_THREADS
    list(_pool.map(part, [jobs[i:i + step] for i in range(0, len(jobs), step)]))

Geometry = Tuple[Tuple[int, int, int, int], Tuple[int, int], Tuple[int, int]]  # source view (i, j, h, w), resized (oh, ow), crop (top, left)


def as_u8_hwc(img):
    """One decoded image -> (uint8 [H, W, 3 or 4] array or CUDA tensor, bytes per pixel).  PIL images go through convert('RGB')
    like the reference (clip_transform.py:27-28); arrays / tensors must already be uint8 HWC with 3 (RGB) or 4 (RGBX) channels."""
    if hasattr(img, "convert") and hasattr(img, "size") and not isinstance(img, (np.ndarray, Tensor)):
        return np.asarray(img.convert("RGB")), 3
    if isinstance(img, Tensor):
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] not in (3, 4):
            raise ops.MmamdError(f"image tensor must be uint8 [H, W, 3|4], got {img.dtype} {tuple(img.shape)}")
        if img.is_cuda:
            if img.stride(2) != 1 or img.stride(1) != img.shape[2]:
                raise ops.MmamdError("CUDA image tensors must be dense along W and C (row stride is free)")
            return img, int(img.shape[2])
        img = img.numpy()
    if isinstance(img, np.ndarray):
        if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] not in (3, 4):
            raise ops.MmamdError(f"image array must be uint8 [H, W, 3|4], got {img.dtype} {img.shape}")
        return np.ascontiguousarray(img), int(img.shape[2])
    raise TypeError(f"unsupported image type {type(img)}")


def random_resized_crop_params(height: int, width: int, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)) -> Tuple[int, int, int, int]:
    """The crop box (top, left, h, w) torchvision's RandomResizedCrop.get_params draws -- same draws from torch's global CPU
    generator in the same order: up to 10 tries of area * U(scale), exp(U(log ratio)), a random origin; then the centred fallback."""
    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))
    for _ in range(10):
        target_area = area * torch.empty(1).uniform_(scale[0], scale[1]).item()
        aspect_ratio = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
        w = int(round(math.sqrt(target_area * aspect_ratio)))
        h = int(round(math.sqrt(target_area / aspect_ratio)))
        if 0 < w <= width and 0 < h <= height:
            i = torch.randint(0, height - h + 1, size=(1,)).item()
            j = torch.randint(0, width - w + 1, size=(1,)).item()
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


@lru_cache(maxsize=4096)
def _geometry(vh: int, vw: int, oh: int, ow: int, top: int, left: int, ch: int, cw: int, filter: str):
    """Tables of one image geometry (a vh x vw source view resized to oh x ow, window ch x cw at (top, left)): the flat int32 block
    [H coefficients | H bounds | V coefficients | V bounds (first row relative to row0)], the four offsets into it, the tap-table
    widths, the source row range the vertical window needs and the source columns per row the horizontal pass reads."""
    kh, bh = axis_tables(vw, ow, left, cw, filter)
    kv, bv = axis_tables(vh, oh, top, ch, filter)
    row0 = int(bv[:, 0].min())
    nrows = int((bv[:, 0] + bv[:, 1]).max()) - row0
    bv = bv - np.array([row0, 0], np.int32)
    parts = [kh.reshape(-1), bh.reshape(-1), kv.reshape(-1), bv.reshape(-1)]
    rel, o = [], 0
    for q in parts:
        rel.append(o)
        o += q.size
    flat = np.concatenate(parts)
    flat.setflags(write=False)
    return flat, tuple(rel), kh.shape[1], kv.shape[1], row0, nrows, int(bh[-1, 0] + bh[-1, 1] - bh[0, 0])


class DeviceResampler:
    """crop_hw: the (h, w) window every image of a batch is resampled into; filter: 'bicubic' | 'lanczos' (Pillow's definitions)."""

    def __init__(self, crop_hw: Tuple[int, int], filter: str = "bicubic", device: Optional[Union[str, torch.device]] = None) -> None:
        self.crop_hw = (int(crop_hw[0]), int(crop_hw[1]))
        self.filter = filter
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self._plans: dict = {}  # (shapes, geometries) of a batch -> its plan; evaluation loaders repeat the same few

    def plan(self, items: Sequence, geoms: Sequence[Geometry]):
        """Host geometry of a batch: the descriptor table (word 0 still relative to each image's first byte), the concatenated
        int32 coefficient tables, each host image's offset in the pixel staging area (None for device tensors), and the sizes
        (pixel staging bytes, tmp bytes, max rows of the vertical window, max source bytes per row of the horizontal pass, max size of
        a horizontal coefficient table).
        Images with the same geometry share one copy of their tables."""
        ch, cw = self.crop_hw
        B = len(items)
        batch_key = (tuple((a.shape, px, a.stride(0) if isinstance(a, Tensor) else -1) for a, px in items), tuple(geoms))
        hit = self._plans.get(batch_key)
        if hit is not None:
            return (hit[0].copy(),) + hit[1:]
        desc = np.zeros((B, _DESC), np.int64)
        tabs, slots, tab_len, host_off, host_len, tmp_len, max_rows, max_seg, max_coef = [], {}, 0, [], 0, 0, 1, 0, 0
        for b, ((a, px), ((vi, vj, vh, vw), (oh, ow), (top, left))) in enumerate(zip(items, geoms)):
            h, w = int(a.shape[0]), int(a.shape[1])
            if h < 1 or w < 1:
                raise ops.MmamdError("empty image")
            if vi < 0 or vj < 0 or vh < 1 or vw < 1 or vi + vh > h or vj + vw > w or top < 0 or left < 0 or top + ch > oh or left + cw > ow:
                raise ops.MmamdError(f"geometry out of range for a {h}x{w} image: view {(vi, vj, vh, vw)}, resized {(oh, ow)}, crop at {(top, left)}")
            key = (vh, vw, oh, ow, top, left, ch, cw, self.filter)
            flat, rel, ksh, ksv, row0, nrows, seg_cols = _geometry(*key)
            base = slots.get(key)
            if base is None:
                base = slots[key] = tab_len
                tabs.append(flat)
                tab_len += flat.size
            stride = a.stride(0) if isinstance(a, Tensor) else w * px
            desc[b] = (vi * stride + vj * px, stride, vh, vw, row0, nrows, base + rel[0], base + rel[1], ksh, base + rel[2],
                       base + rel[3], ksv, tmp_len, px, 0, 0)
            tmp_len += (nrows * cw * 3 + 15) // 16 * 16
            max_rows = max(max_rows, nrows)
            max_seg = max(max_seg, seg_cols * px)
            max_coef = max(max_coef, cw * ksh)
            if isinstance(a, Tensor):
                host_off.append(None)
            else:
                host_off.append(host_len)
                host_len += (a.size + 15) // 16 * 16
        tables = np.concatenate(tabs) if tabs else np.zeros(0, np.int32)
        plan = (desc, tables, host_off, host_len, tmp_len, max_rows, max_seg, max_coef)
        if len(self._plans) >= 8:
            self._plans.pop(next(iter(self._plans)))
        self._plans[batch_key] = (desc.copy(),) + plan[1:]
        return plan

    def upload(self, items: Sequence) -> List:
        """Move the host images of `items` to the device in one pinned-buffer copy; device tensors pass through.  Lets several
        resampling runs (FLAVA: encoder image + codebook image) read the same source pixels."""
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise ops.MmamdError(f"image transforms run on a HIP device (device={self.device}, available="
                                 f"{torch.cuda.is_available()}): there is no CPU path")
        offs, total = [], 0
        for a, _ in items:
            offs.append(None if isinstance(a, Tensor) else total)
            if not isinstance(a, Tensor):
                total += (a.size + 15) // 16 * 16
        if total == 0:
            return list(items)
        stage = torch.empty(total, dtype=torch.uint8, pin_memory=True)
        sn = stage.numpy()
        _stage_pixels(sn, [(o, a) for (a, _), o in zip(items, offs) if o is not None])
        dev = stage.to(self.device, non_blocking=True)
        return [(a, px) if o is None else (dev[o:o + a.size].view(a.shape), px) for (a, px), o in zip(items, offs)]

    def run(self, items: Sequence, geoms: Sequence[Geometry], lut: Optional[np.ndarray], want_f32: bool, patch: int = 0, kpad: int = 0,
            want_u8: bool = False):
        """-> (float32 [B,3,ch,cw] | None, bf16 patch rows | None, uint8 [B,ch,cw,3] | None) on the device, stream-ordered."""
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise ops.MmamdError(f"image transforms run on a HIP device (device={self.device}, available="
                                 f"{torch.cuda.is_available()}): there is no CPU path")
        ch, cw = self.crop_hw
        B = len(items)
        if B == 0:  # nothing to launch: empty outputs of the right shapes
            k = kpad or 3 * patch * patch
            return (torch.empty((0, 3, ch, cw), dtype=torch.float32, device=self.device) if want_f32 else None,
                    torch.empty((0, k), dtype=torch.bfloat16, device=self.device) if patch else None,
                    torch.empty((0, ch, cw, 3), dtype=torch.uint8, device=self.device) if want_u8 else None)
        desc, tables, host_off, host_len, tmp_len, max_rows, max_seg, max_coef = self.plan(items, geoms)
        # one staging buffer [desc | value table | coefficient tables | pixels of the host images], one H2D copy
        o_lut = B * _DESC * 8
        o_tab = o_lut + 768 * 4
        n_tab = tables.size * 4
        o_pix = (o_tab + n_tab + 15) // 16 * 16
        stage = torch.empty(o_pix + host_len, dtype=torch.uint8, pin_memory=True)
        dev = torch.empty(stage.numel(), dtype=torch.uint8, device=self.device)
        sn = stage.numpy()
        jobs = []
        for b, (a, px) in enumerate(items):
            if host_off[b] is None:
                desc[b, 0] += a.data_ptr()
            else:
                o = o_pix + host_off[b]
                jobs.append((o, a))
                desc[b, 0] += dev.data_ptr() + o
        _stage_pixels(sn, jobs)
        sn[:o_lut] = desc.reshape(-1).view(np.uint8)
        if lut is not None:
            sn[o_lut:o_tab] = np.ascontiguousarray(lut, np.float32).reshape(-1).view(np.uint8)
        sn[o_tab:o_tab + n_tab] = tables.view(np.uint8)
        dev.copy_(stage, non_blocking=True)
        tmp = torch.empty(max(tmp_len, 16), dtype=torch.uint8, device=self.device)
        tab_dev = dev[o_tab:o_tab + max(n_tab, 4)].view(torch.int32)
        lut_dev = dev[o_lut:o_tab].view(torch.float32) if lut is not None else None
        return ops.image_resample(dev[:o_lut].view(torch.int64), tab_dev, tmp, B, ch, cw, max_rows, max_seg, max_coef, lut_dev, want_f32,
                                  patch, kpad, want_u8)
