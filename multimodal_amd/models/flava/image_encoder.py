"""Host-side mirror of torchmultimodal/models/flava/image_encoder.py (PatchEmbeddings :27-66, ImageEmbeddings :69-177,
ImageTransformer :180-237, flava_image_encoder :240-285).

The patch embedding Conv2d(kernel = stride = patch) is im2col (csrc/rowops.hip: patchify_kernel, bf16) + one MFMA GEMM
with the conv bias in the epilogue; mask-token blending, the CLS row and the position embeddings are one row kernel
(flava_image_embed_kernel).
"""
from __future__ import annotations

import math
import warnings
from functools import partial
from typing import Any, Callable, Optional, Tuple

import torch
from torch import nn, Tensor

from ... import ops
from ..._packing import PackedCache
from ...modules.layers.normalizations import Fp32LayerNorm
from ...modules.layers.transformer import TransformerOutput
from ...schedule import get_schedule
from ..._autograd import wants_grad
from ...modules.losses.flava import Pooler
from .transformer import init_transformer_weights, TransformerEncoder


def to_2tuple(x: int) -> Tuple[int, int]:
    return (x, x)


class PatchEmbeddings(nn.Module):
    """Image to Patch Embedding."""

    def __init__(self, image_size: int = 224, patch_size: int = 16, num_channels: int = 3, embed_dim: int = 768) -> None:
        super().__init__()
        image_size = to_2tuple(image_size)
        patch_size = to_2tuple(patch_size)
        self.image_size = image_size
        self.patch_size = patch_size
        self.num_patches = (image_size[1] // patch_size[1]) * (image_size[0] // patch_size[0])
        self.projection = nn.Conv2d(num_channels, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self._packed = PackedCache()

    def forward(self, pixel_values: Tensor, interpolate_pos_encoding: bool = False) -> Tensor:
        _, _, height, width = pixel_values.shape
        if not interpolate_pos_encoding:
            if height != self.image_size[0] or width != self.image_size[1]:
                raise ValueError(
                    f"Input image size ({height}*{width}) doesn't match model ({self.image_size[0]}*{self.image_size[1]}).")
        elif height != width or height % self.patch_size[0] != 0 or self.patch_size[0] != self.patch_size[1]:
            raise ops.MmamdError(f"interpolate_pos_encoding on the MI355X path takes square images whose side is a multiple of the "
                                 f"patch size (got {height}x{width}, patch {self.patch_size[0]}): the patch gather is written for square grids")
        B, C = pixel_values.shape[:2]
        if C != self.projection.in_channels:
            raise ValueError(f"expected {self.projection.in_channels} channels, got {C}")
        P = self.patch_size[0]
        k = C * P * P
        kpad = (k + 63) // 64 * 64
        w = self.projection.weight
        wk = self._packed.get(w, torch.bfloat16).view(w.shape[0], k)
        if kpad != k:
            raise ops.MmamdError(f"patch embedding: C*P*P = {k} must be a multiple of 64 on the MI355X path")
        px = pixel_values if pixel_values.is_contiguous() else pixel_values.contiguous()
        cols = ops.patchify(px, P, kpad)
        bias = self._packed.get(self.projection.bias, torch.float32) if self.projection.bias is not None else None
        x = ops.gemm_bf16(cols, wk, bias, out_dtype=torch.float32)
        return x.view(B, (height // P) * (width // P), -1)


class ImageEmbeddings(nn.Module):
    """Construct the CLS token, position and patch embeddings."""

    def __init__(self, image_size: int = 224, patch_size: int = 16, num_channels: int = 3, hidden_size: int = 768,
                 hidden_dropout_prob: float = 0.0, use_image_masking: bool = True) -> None:
        super().__init__()
        self.cls_token = nn.Parameter(torch.zeros(1, 1, hidden_size))
        self.patch_embeddings = PatchEmbeddings(image_size=image_size, patch_size=patch_size, num_channels=num_channels,
                                                embed_dim=hidden_size)
        num_patches = self.patch_embeddings.num_patches
        self.position_embeddings = nn.Parameter(torch.zeros(1, num_patches + 1, hidden_size))
        self.dropout = nn.Dropout(hidden_dropout_prob)
        if use_image_masking:
            self.mask_token = nn.Parameter(torch.zeros(1, 1, hidden_size))
        else:
            self.mask_token = None
        self._packed = PackedCache()

    def interpolate_pos_encoding(self, embeddings: Tensor, height: int, width: int) -> Tensor:
        """Position table for a different resolution (reference :102-137): `embeddings` is the [B, 1 + npatch, d] sequence (CLS
        included) and only supplies npatch.  Returns [1, 1 + npatch, d]; the trained table itself when nothing changes."""
        return self._interp_table(embeddings.shape[1] - 1, height, width)

    def _interp_table(self, npatch: int, height: int, width: int) -> Tensor:
        n = self.position_embeddings.shape[1] - 1
        if npatch == n and height == width:
            return self.position_embeddings
        h0 = height // self.patch_embeddings.patch_size[0] + 0.1  # reference :118-121: +0.1 against floor() of the scaled size
        w0 = width // self.patch_embeddings.patch_size[1] + 0.1
        table = self._packed.get(self.position_embeddings, torch.float32).view(n + 1, -1)
        out = ops.bicubic_pos_embed(table, int(h0), int(w0), h0 / math.sqrt(n), w0 / math.sqrt(n))  # mmamd_bicubic_pos_embed
        return out.view(1, out.shape[0], out.shape[1])

    def forward(self, pixel_values: Tensor, image_patches_mask: Optional[Tensor] = None,
                interpolate_pos_encoding: bool = False) -> Tensor:
        B = pixel_values.shape[0]
        if wants_grad(self, pixel_values):
            if image_patches_mask is not None and self.mask_token is None:
                warnings.warn("image_patches_mask passed but use_image_masking in init was false. Ignoring.")
                image_patches_mask = None
            pe_mod = self.patch_embeddings
            H, W = pixel_values.shape[2], pixel_values.shape[3]
            from ._train import BicubicTableFn, FlavaImageEmbedFn

            from ..._autograd import dropout_train

            pos = self.position_embeddings
            if interpolate_pos_encoding:  # (reference :170-173) another resolution: the table resampled bicubically, differentiably (r05)
                p0, p1 = pe_mod.patch_size
                if H != W or p0 != p1 or H % p0 != 0:
                    raise ops.MmamdError("interpolate_pos_encoding on the MI355X path: square images whose side is a multiple of the patch size")
                n = pos.shape[1] - 1
                if (H // p0) * (W // p1) != n:
                    h0, w0 = H // p0 + 0.1, W // p1 + 0.1  # reference :118-121
                    pos = BicubicTableFn.apply(pos, int(h0), int(w0), h0 / math.sqrt(n), w0 / math.sqrt(n))
            elif H != pe_mod.image_size[0] or W != pe_mod.image_size[1]:
                raise ValueError(f"Input image size ({H}*{W}) doesn't match model ({pe_mod.image_size[0]}*{pe_mod.image_size[1]}).")
            emb = FlavaImageEmbedFn.apply(pixel_values, pe_mod.projection.weight, pe_mod.projection.bias, self.cls_token,
                                          pos, pe_mod.patch_size[0], image_patches_mask,
                                          self.mask_token if image_patches_mask is not None else None)
            return dropout_train(emb, self.dropout.p)  # reference flava/image_encoder.py:165: dropout on the assembled embeddings
        if self.training and self.dropout.p > 0:
            raise ops.MmamdError("embedding dropout applies on the differentiable (train mode, grad enabled) forward only: call .eval() for inference")
        pe = self.patch_embeddings(pixel_values, interpolate_pos_encoding=interpolate_pos_encoding)
        G2 = pe.shape[1]
        pk, f32 = self._packed.get, torch.float32
        mask, mask_token = None, None
        if image_patches_mask is not None:
            if self.mask_token is not None:
                m = image_patches_mask.reshape(B, -1)
                if m.dtype != torch.int64:
                    m = m.to(torch.int64)  # bool / int masks: a dtype cast of B*G2 flags, not arithmetic
                mask, mask_token = m.contiguous(), pk(self.mask_token, f32)
            else:
                warnings.warn("image_patches_mask passed but use_image_masking in init was false. Ignoring.")
        pos = pk(self.position_embeddings, f32)
        if interpolate_pos_encoding:
            pos = self._interp_table(G2, pixel_values.shape[2], pixel_values.shape[3]).contiguous()
        x = ops.flava_image_embed(pe.view(B * G2, -1), pk(self.cls_token, f32), pos, B, G2, mask, mask_token)
        return x.view(B, G2 + 1, -1)


class ImageTransformer(nn.Module):
    def __init__(self, embeddings: nn.Module, encoder: nn.Module, layernorm: nn.Module, pooler: nn.Module,
                 weight_init_fn: Optional[Callable] = None, initializer_range: float = 0.02, **kwargs: Any) -> None:
        super().__init__()
        self.embeddings = embeddings
        self.encoder = encoder
        self.layernorm = layernorm
        self.pooler = pooler
        if weight_init_fn is None:
            weight_init_fn = partial(init_transformer_weights, initializer_range=initializer_range)
        self.apply(weight_init_fn)

    def forward(self, pixel_values: Optional[Tensor] = None, image_patches_mask: Optional[Tensor] = None,
                attention_mask: Optional[Tensor] = None) -> TransformerOutput:
        if pixel_values is None:
            raise ValueError("You have to specify pixel_values")
        embedding_output = self.embeddings(pixel_values, image_patches_mask=image_patches_mask)
        encoder_output = self.encoder(embedding_output, attention_mask=attention_mask, return_attn_weights=get_schedule().flava_attentions,
                                      return_hidden_states=True)
        sequence_output = self.layernorm(encoder_output.last_hidden_state)
        pooled_output = self.pooler(sequence_output) if self.pooler is not None else None
        return TransformerOutput(last_hidden_state=sequence_output, pooler_output=pooled_output,
                                 hidden_states=encoder_output.hidden_states, attentions=encoder_output.attentions)


def flava_image_encoder(
    hidden_size: int = 768,
    num_attention_heads: int = 12,
    num_hidden_layers: int = 12,
    use_image_masking: bool = False,
    dropout: float = 0.0,
    intermediate_size: int = 3072,
    intermediate_activation: Callable[..., nn.Module] = nn.GELU,
    layer_norm_eps: float = 1e-12,
    image_size: int = 224,
    patch_size: int = 16,
    num_channels: int = 3,
) -> ImageTransformer:
    embeddings = ImageEmbeddings(image_size=image_size, patch_size=patch_size, num_channels=num_channels,
                                 hidden_size=hidden_size, hidden_dropout_prob=dropout, use_image_masking=use_image_masking)
    encoder = TransformerEncoder(n_layer=num_hidden_layers, d_model=hidden_size, n_head=num_attention_heads,
                                 dim_feedforward=intermediate_size, activation=intermediate_activation,
                                 layer_norm_eps=layer_norm_eps, dropout=dropout, norm_first=True)
    layernorm = Fp32LayerNorm(hidden_size, eps=layer_norm_eps)
    pooler = Pooler(hidden_size=hidden_size)
    return ImageTransformer(embeddings=embeddings, encoder=encoder, layernorm=layernorm, pooler=pooler)
