"""Host-side mirror of torchmultimodal/modules/layers/patch_embedding.py:17-152 (PatchEmbeddings, PatchEmbeddingsOutput).
Conv2d(kernel = stride = patch) = im2col (patchify_kernel) + one MFMA GEMM with the conv bias; mask-token blend, optional CLS
row and position embeddings are one row kernel (flava_image_embed_kernel)."""

import math
import warnings
from typing import NamedTuple, Optional, Tuple, Union

import torch
from torch import nn, Tensor

from ... import _torch_ops, ops
from ..._packing import PackedCache

_torch_ops.try_load()


class PatchEmbeddingsOutput(NamedTuple):
    embeddings: Tensor
    random_mask: Optional[Tensor] = None
    ids_restore: Optional[Tensor] = None


class PatchEmbeddings(nn.Module):
    def __init__(self, image_size: Union[int, Tuple[int, int]] = 224, patch_size: int = 16, num_channels: int = 3,
                 hidden_size: int = 768, hidden_dropout_prob: float = 0.0, use_image_masking: bool = False,
                 patch_drop_rate: Optional[Union[float, Tuple[float, float]]] = None, include_cls_embed: bool = True) -> None:
        super().__init__()
        if isinstance(image_size, int):
            image_size = (image_size, image_size)
        if image_size[0] % patch_size != 0 or image_size[1] % patch_size != 0:
            raise ValueError("Image size needs to be divisible by patch size")
        self.num_patches_h = image_size[0] // patch_size
        self.num_patches_w = image_size[1] // patch_size
        num_patches = self.num_patches_h * self.num_patches_w
        self.include_cls_embed = include_cls_embed
        if self.include_cls_embed:
            self.cls_token = nn.Parameter(torch.zeros(1, 1, hidden_size))
            num_patches = num_patches + 1
        self.conv_projection = nn.Conv2d(num_channels, hidden_size, kernel_size=patch_size, stride=patch_size)
        self._init_conv_weights()
        self.image_size: Tuple[int, int] = image_size
        self.position_embeddings = nn.Parameter(torch.zeros(1, num_patches, hidden_size))
        self.dropout = nn.Dropout(hidden_dropout_prob)
        if use_image_masking:
            self.mask_token = nn.Parameter(torch.zeros(1, 1, hidden_size))
        else:
            self.mask_token = None
        self.patch_drop_rate = patch_drop_rate
        self._patch: int = patch_size
        self._packed = PackedCache()

    def _init_conv_weights(self) -> None:
        fan_in = self.conv_projection.in_channels * self.conv_projection.kernel_size[0] * self.conv_projection.kernel_size[1]
        nn.init.trunc_normal_(self.conv_projection.weight, std=math.sqrt(1 / fan_in))
        assert self.conv_projection.bias is not None
        nn.init.zeros_(self.conv_projection.bias)

    def forward(self, pixel_values: Tensor, image_patches_mask: Optional[Tensor] = None) -> PatchEmbeddingsOutput:
        if torch.jit.is_scripting():
            return self._forward_ops(pixel_values, image_patches_mask)
        else:
            return self._forward_host(pixel_values, image_patches_mask)

    def _forward_ops(self, pixel_values: Tensor, image_patches_mask: Optional[Tensor]) -> PatchEmbeddingsOutput:
        """The forward through the dispatcher ops (torch.ops.mmamd.image_embed: im2col + GEMM with the conv bias + CLS row + position
        embeddings) — what torch.jit.script / torch.compile see.  Inference, no patch masking."""
        if image_patches_mask is not None:
            raise RuntimeError("scripted PatchEmbeddings on the MI355X path takes no image_patches_mask (use the eager forward)")
        if pixel_values.dim() != 4 or pixel_values.size(2) != self.image_size[0] or pixel_values.size(3) != self.image_size[1]:
            raise ValueError("Input image size doesn't match the image size expected by model")
        if self.image_size[0] != self.image_size[1]:
            raise RuntimeError("non-square images are not implemented on the MI355X path")
        B = pixel_values.size(0)
        cls: Optional[Tensor] = None
        if hasattr(self, "cls_token"):  # (resolved when the module is scripted: the parameter exists only with include_cls_embed)
            cls = self.cls_token
        conv_bias = self.conv_projection.bias
        assert conv_bias is not None
        x = torch.ops.mmamd.image_embed(pixel_values.contiguous(), self.conv_projection.weight, conv_bias, cls, self.position_embeddings,
                                        self._patch)
        return PatchEmbeddingsOutput(embeddings=x.view(B, -1, x.size(1)))

    @torch.jit.unused
    def _forward_host(self, pixel_values: Tensor, image_patches_mask: Optional[Tensor] = None) -> PatchEmbeddingsOutput:
        if torch.compiler.is_compiling() and image_patches_mask is None and not (self.training and torch.is_grad_enabled()):
            return self._forward_ops(pixel_values, image_patches_mask)
        batch_size, num_channels, height, width = pixel_values.shape
        if height != self.image_size[0] or width != self.image_size[1]:
            raise ValueError(f"Input image size ({height}*{width}) doesn't match image size "
                             f"{self.image_size[0]}*{self.image_size[1]} expected by model")
        if height != width:
            raise ops.MmamdError("non-square images are not implemented on the MI355X path")
        if self.training and self.patch_drop_rate is not None:
            raise ops.MmamdError("patch dropping (patch_drop_rate) in training mode is not implemented on the MI355X path")
        P = self.conv_projection.kernel_size[0]
        if self.training and torch.is_grad_enabled() and self.conv_projection.weight.requires_grad:
            from ...models.flava._train import FlavaImageEmbedFn  # differentiable path (conv + bias, optional CLS / mask token, + pos)

            if image_patches_mask is not None and self.mask_token is None:
                warnings.warn("image_patches_mask passed but use_image_masking in init was false. Ignoring.")
                image_patches_mask = None
            from ..._autograd import dropout_train

            x = FlavaImageEmbedFn.apply(pixel_values, self.conv_projection.weight, self.conv_projection.bias,
                                        self.cls_token if self.include_cls_embed else None, self.position_embeddings, P,
                                        image_patches_mask, self.mask_token if image_patches_mask is not None else None)
            return PatchEmbeddingsOutput(embeddings=dropout_train(x, self.dropout.p))  # reference :150: dropout on the assembled embeddings
        if self.training and self.dropout.p > 0:
            raise ops.MmamdError("embedding dropout applies on the differentiable (train mode, grad enabled) forward only: call .eval() for inference")
        w = self.conv_projection.weight
        k = num_channels * P * P
        kpad = (k + 63) // 64 * 64
        pk, bf, f32 = self._packed.get, torch.bfloat16, torch.float32
        if kpad == k:
            wk = pk(w, bf).view(w.shape[0], k)
        else:  # e.g. patch 14: 588 -> 640, zero columns
            wk = self._padded_weight(w, k, kpad)
        px = pixel_values if pixel_values.is_contiguous() else pixel_values.contiguous()
        cols = ops.patchify(px, P, kpad)
        pe = ops.gemm_bf16(cols, wk, pk(self.conv_projection.bias, f32), out_dtype=f32)
        G2 = self.num_patches_h * self.num_patches_w
        mask, mask_token = None, None
        if image_patches_mask is not None:
            if self.mask_token is not None:
                m = image_patches_mask.reshape(batch_size, -1)
                mask = (m if m.dtype == torch.int64 else m.to(torch.int64)).contiguous()
                mask_token = pk(self.mask_token, f32)
            else:
                warnings.warn("image_patches_mask passed but use_image_masking in init was false. Ignoring.")
        cls = pk(self.cls_token, f32) if self.include_cls_embed else None
        x = ops.flava_image_embed(pe, cls, pk(self.position_embeddings, f32), batch_size, G2, mask, mask_token)
        return PatchEmbeddingsOutput(embeddings=x.view(batch_size, G2 + (1 if cls is not None else 0), -1))

    def _padded_weight(self, w: Tensor, k: int, kpad: int) -> Tensor:
        """[E, C, P, P] conv weight as a bf16 [E, kpad] GEMM weight with zero columns k..kpad (cached)."""
        key = (w.data_ptr(), w._version, kpad)
        hit = getattr(self, "_wpad", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        buf = torch.zeros((w.shape[0], kpad), dtype=torch.bfloat16, device=w.device)  # memset
        buf[:, :k].copy_(ops.convert(w.detach().contiguous().view(w.shape[0], k), torch.bfloat16))  # strided placement copy
        self._wpad = (key, buf)
        return buf
