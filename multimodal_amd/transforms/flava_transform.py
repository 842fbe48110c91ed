"""Host-side mirror of torchmultimodal/transforms/flava_transform.py: map_pixels, ImageMaskingGenerator,
FLAVAImageTransform.  The image work runs on the device through mmamd_image_resample (transforms/_device_resample.py):

  eval   TwoWayResize (flava_transform.py:109-150): the image resized to exactly (S, S) with bicubic -> encoder image; THAT resized
         uint8 image resized again to (C, C) with Lanczos -> codebook image.  Two launches, the second reads the first's uint8
         output on the device.
  train  TwoWayRandomResizedCrop (:153-209): one crop box (torchvision RandomResizedCrop.get_params, scale (0.9, 1)), resized to
         (S, S) bicubic and to (C, C) Lanczos, both from the original pixels (uploaded once).
  then   ToTensor + Normalize for the encoder image, ToTensor + map_pixels for the codebook image -- as byte -> float value tables.

The block-wise mask generator (BEiT) is host logic on Python's `random`, draw for draw the reference's (fixture
tests/golden/flava_transform.npz was produced by the reference class under fixed seeds)."""
from __future__ import annotations

import math
import random
from typing import Dict, List, Mapping, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from torch import Tensor

from .. import ops
from ._device_resample import DeviceResampler, as_u8_hwc, random_resized_crop_params
from ._resample import map_pixels_lut, normalize_lut

IMAGE_PRETRAINING_MEAN = (0.48145466, 0.4578275, 0.40821073)
IMAGE_PRETRAINING_STD = (0.26862954, 0.26130258, 0.27577711)
LOGIT_LAPLACE_EPS: float = 0.1


def map_pixels(x: Tensor) -> Tensor:
    """flava_transform.py:24-28 -- a public helper of the reference, kept for API parity (plain tensor arithmetic wherever `x`
    lives); FLAVAImageTransform itself never calls it: the codebook image comes out of the resampling kernel already mapped."""
    if x.dtype != torch.float:
        raise ValueError("expected input to have type float")
    return (1 - 2 * LOGIT_LAPLACE_EPS) * x + LOGIT_LAPLACE_EPS


class ImageMaskingGenerator:
    """Block-wise patch masking of BEiT (flava_transform.py:31-106): rectangles of random area / aspect ratio are OR-ed into a
    height x width grid until num_masking_patches cells are set (or 10 draws in a row add nothing)."""

    def __init__(self, input_size: Union[Tuple[int, int], int], num_masking_patches: int, min_num_patches: int = 4,
                 max_num_patches: Optional[int] = None, min_aspect: float = 0.3, max_aspect: Optional[float] = None) -> None:
        if not isinstance(input_size, tuple):
            input_size = (input_size,) * 2
        self.height, self.width = input_size
        self.num_patches = self.height * self.width
        self.num_masking_patches = num_masking_patches
        self.min_num_patches = min_num_patches
        self.max_num_patches = num_masking_patches if max_num_patches is None else max_num_patches
        max_aspect = max_aspect or 1 / min_aspect
        self.log_aspect_ratio = (math.log(min_aspect), math.log(max_aspect))

    def __repr__(self) -> str:
        return "Generator(%d, %d -> [%d ~ %d], max = %d, %.3f ~ %.3f)" % (
            self.height, self.width, self.min_num_patches, self.max_num_patches, self.num_masking_patches,
            self.log_aspect_ratio[0], self.log_aspect_ratio[1])

    def get_shape(self) -> Tuple[int, int]:
        return self.height, self.width

    def _mask(self, mask: np.ndarray, max_mask_patches: int) -> int:
        """One rectangle: up to 10 draws; a draw that fits the grid and adds between 1 and max_mask_patches new cells is applied."""
        for _ in range(10):
            target_area = random.uniform(self.min_num_patches, max_mask_patches)
            aspect_ratio = math.exp(random.uniform(*self.log_aspect_ratio))
            h = int(round(math.sqrt(target_area * aspect_ratio)))
            w = int(round(math.sqrt(target_area / aspect_ratio)))
            if w < self.width and h < self.height:
                top = random.randint(0, self.height - h)
                left = random.randint(0, self.width - w)
                box = mask[top:top + h, left:left + w]
                fresh = h * w - int(box.sum())
                if 0 < fresh <= max_mask_patches:
                    box[...] = 1
                    return fresh
        return 0

    def __call__(self) -> np.ndarray:
        mask = np.zeros(shape=self.get_shape(), dtype=np.int64)
        count = 0
        while count < self.num_masking_patches:
            delta = self._mask(mask, min(self.num_masking_patches - count, self.max_num_patches))
            if delta == 0:
                break
            count += delta
        return mask


def _filter_name(mode) -> str:
    name = str(getattr(mode, "value", mode)).lower()
    if name not in ("bicubic", "lanczos"):
        raise ops.MmamdError(f"interpolation {mode!r} is not implemented on the MI355X path (bicubic / lanczos)")
    return name


def _pair(size) -> Tuple[int, int]:
    return (int(size), int(size)) if not isinstance(size, (list, tuple)) else (int(size[0]), int(size[1]))


class FLAVAImageTransform:
    """FLAVA image transform (flava_transform.py:212-314): resize / random resized crop to the encoder size and to the codebook
    size, normalisation, BEiT patch mask.  Same constructor and call contract; `device` and `batch()` are extensions.

    __call__(image) -> {"image": f32 [3,S,S], "image_for_codebook": f32 [3,C,C], "image_patches_mask": int64 [W,W]};
    __call__([images]) -> the same keys with lists of per-image tensors.  Image tensors live on `device`, masks on the host, like
    the reference's.  batch([images]) -> stacked device tensors ([B,3,S,S], [B,3,C,C], [B,W,W]) for the training loop."""

    def __init__(self, is_train: bool = True, encoder_input_size: int = 224, codebook_input_size: int = 112,
                 scale: Tuple[float, float] = (0.9, 1.0), encoder_interpolation="bicubic", codebook_interpolation="lanczos",
                 image_mean: Tuple[float, float, float] = IMAGE_PRETRAINING_MEAN, image_std: Tuple[float, float, float] = IMAGE_PRETRAINING_STD,
                 mask_window_size: int = 14, mask_num_patches: int = 75, mask_max_patches: Optional[int] = None,
                 mask_min_patches: int = 16, device: Optional[Union[str, torch.device]] = None) -> None:
        self.is_train = is_train
        self.scale = scale
        self.encoder_hw = _pair(encoder_input_size)
        self.codebook_hw = _pair(codebook_input_size)
        self.encoder = DeviceResampler(self.encoder_hw, _filter_name(encoder_interpolation), device)
        self.codebook = DeviceResampler(self.codebook_hw, _filter_name(codebook_interpolation), device)
        self.image_lut = normalize_lut([float(v) for v in image_mean], [float(v) for v in image_std])
        self.codebook_lut = map_pixels_lut(LOGIT_LAPLACE_EPS)
        self.masked_position_generator = ImageMaskingGenerator(mask_window_size, num_masking_patches=mask_num_patches,
                                                               max_num_patches=mask_max_patches, min_num_patches=mask_min_patches)

    def _images(self, images: Sequence) -> Tuple[Tensor, Tensor]:
        items = [as_u8_hwc(im) for im in images]
        eh, ew = self.encoder_hw
        chh, cww = self.codebook_hw
        if self.is_train:
            views = [random_resized_crop_params(int(a.shape[0]), int(a.shape[1]), scale=self.scale) for a, _ in items]
            items = self.encoder.upload(items)  # both resamplings read the original pixels
            enc = self.encoder.run(items, [(v, (eh, ew), (0, 0)) for v in views], self.image_lut, True)[0]
            cb = self.codebook.run(items, [(v, (chh, cww), (0, 0)) for v in views], self.codebook_lut, True)[0]
            return enc, cb
        whole = [((0, 0, int(a.shape[0]), int(a.shape[1])), (eh, ew), (0, 0)) for a, _ in items]
        enc, _, small = self.encoder.run(items, whole, self.image_lut, True, 0, 0, True)
        second = [(small[b], 3) for b in range(len(items))]  # the codebook image is a resize of the RESIZED image (:139-149)
        cb = self.codebook.run(second, [((0, 0, eh, ew), (chh, cww), (0, 0))] * len(items), self.codebook_lut, True)[0]
        return enc, cb

    def batch(self, images: Sequence) -> Dict[str, Tensor]:
        enc, cb = self._images(list(images))
        masks = np.stack([self.masked_position_generator() for _ in range(enc.shape[0])]) if enc.shape[0] else np.zeros(
            (0,) + self.masked_position_generator.get_shape(), np.int64)
        return {"image": enc, "image_for_codebook": cb,
                "image_patches_mask": torch.from_numpy(masks).to(enc.device, non_blocking=True)}

    def transform(self, image) -> Dict[str, Tensor]:
        enc, cb = self._images([image])
        return {"image": enc[0], "image_for_codebook": cb[0],
                "image_patches_mask": torch.from_numpy(self.masked_position_generator())}

    def __call__(self, images) -> Mapping[str, Union[Tensor, List[Tensor]]]:
        if not isinstance(images, list):
            return self.transform(images)
        enc, cb = self._images(images)
        return {"image": list(enc.unbind(0)), "image_for_codebook": list(cb.unbind(0)),
                "image_patches_mask": [torch.from_numpy(self.masked_position_generator()) for _ in images]}
