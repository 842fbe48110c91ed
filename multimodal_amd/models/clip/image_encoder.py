"""CLIPViTEncoder — host-side mirror of torchmultimodal/models/clip/image_encoder.py:22-113 on the MI355X kernels.

Same constructor signature, same parameter names/shapes (state_dict-compatible), same
ValueErrors; forward() = patchify -> patch-embed GEMM -> (+CLS,+pos,ln_pre) -> transformer stack ->
(ln_post(CLS) @ projection), every step a libmmamd.so kernel.  The ResNet towers of the reference
(image_encoder.py:116-339) are outside the hot path and are not provided.
"""
from __future__ import annotations

import torch
from torch import nn, Tensor

from ... import ops
from ..._packing import PackedCache, PackedModeMixin
from ...modules.layers.normalizations import Fp32LayerNorm
from . import _train
from ._transformer import TransformerStack, forbid_training_forward

EXPANSION = 4


class CLIPViTEncoder(PackedModeMixin, nn.Module):
    """
    Vision transformer encoder for CLIP.

    Args:
        embedding_dim (int): Embedding dimension for text and image encoders projections.
        patch_size (int): The dimension of each patch
        image_size(int): The size (width==height) of input image
        width (int): Dimensionality of the encoder layers and the pooler layer
        heads (int): Number of attention heads for each attention layer in the Transformer encoder
        layers (int): Number of hidden layers in the Transformer encoder

    Inputs:
        x (Tensor): image tensor with dimensions B x C(3) x image_size x image_size
    """

    def __init__(self, embedding_dim: int, patch_size: int, image_size: int, width: int, heads: int, layers: int):
        super().__init__()
        torch._C._log_api_usage_once(f"torchmultimodal.{self.__class__.__name__}")
        # parameter creation order == reference order (image_encoder.py:50-80) so seeded inits coincide
        self.conv = nn.Conv2d(in_channels=3, out_channels=width, kernel_size=patch_size, stride=patch_size, bias=False)
        self.image_size = image_size
        self.patch_size = patch_size
        self.width = width
        self.embedding_dim = embedding_dim

        scale = width**-0.5
        self.cls_token_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((image_size // patch_size) ** 2 + 1, width))
        self.ln_pre = Fp32LayerNorm(width)
        self.encoder = TransformerStack(d_model=width, nhead=heads, dim_feedforward=EXPANSION * width, num_layers=layers)
        self.ln_post = Fp32LayerNorm(width)
        self.projection = nn.Parameter(scale * torch.randn(width, embedding_dim))
        self._packed = PackedCache()
        self._conv_w_cache = None  # (ptr, version, device, packed [width, Kpad] bf16, invalidate_packed() epoch)

    @torch.jit.unused
    def _conv_weight_bf16(self, kpad: int) -> Tensor:
        """conv.weight [w,3,p,p] viewed as the GEMM weight [w, 3*p*p] (K-contiguous), zero-padded to Kpad."""
        w = self.conv.weight.detach()
        K = w.shape[1] * w.shape[2] * w.shape[3]
        if K == kpad:
            return self._packed.get(self.conv.weight, torch.bfloat16).view(w.shape[0], K)
        from ..._packing import packed_epoch

        c = self._conv_w_cache
        if c is not None and c[0] == w.data_ptr() and c[1] == self.conv.weight._version and c[2] == w.device and c[4] == packed_epoch():
            return c[3]
        packed = torch.zeros((w.shape[0], kpad), dtype=torch.bfloat16, device=w.device)  # one-time pack
        packed[:, :K].copy_(self._packed.get(self.conv.weight, torch.bfloat16).view(w.shape[0], K))
        self._conv_w_cache = (w.data_ptr(), self.conv.weight._version, w.device, packed, packed_epoch())
        return packed

    def forward(self, x: Tensor) -> Tensor:
        if x.size(2) != self.image_size or x.size(3) != self.image_size:
            raise ValueError(
                f"Expected input with width and height as {self.image_size}, found {x.size(2)} by {x.size(3)} ")
        if x.size(1) != 3:
            raise ValueError(f"Expected 3 channels found {x.size(1)}")
        if torch.jit.is_scripting():
            return self._forward_ops(x)
        else:
            return self._forward_host(x)

    def _forward_ops(self, x: Tensor) -> Tensor:
        """The forward through the dispatcher ops (torch.ops.mmamd.*, csrc/torch_ops.cpp): scriptable / compilable, inference only."""
        B = x.size(0)
        g = self.image_size // self.patch_size
        S = g * g + 1
        h = torch.ops.mmamd.patch_embed(x.contiguous(), self.conv.weight, self.cls_token_embedding, self.positional_embedding,
                                        self.ln_pre.weight, self.ln_pre.bias, self.ln_pre.eps, self.patch_size)
        h = self.encoder(h, B, S, False)
        return torch.ops.mmamd.pool_proj_normalize(h, B, S, None, self.ln_post.weight, self.ln_post.bias, self.ln_post.eps,
                                                   self.projection, False, False)

    @torch.jit.unused
    def _forward_host(self, x: Tensor) -> Tensor:
        if torch.compiler.is_compiling() and not _train.wants_grad(self, x):
            return self._forward_ops(x)
        if _train.wants_grad(self, x):
            if x.requires_grad:
                # the reference back-propagates into the pixels (Conv2d input gradient); the MI355X training nodes stop at the patch
                # embedding's weight: refuse rather than hand back an image whose .grad silently stays None
                raise NotImplementedError("CLIPViTEncoder: the gradient with respect to the input image is not implemented on the "
                                          "MI355X path (parameters are differentiable; detach the image or run under torch.no_grad())")
            return self._forward_train(x)
        # K1: patch embedding = GEMM over non-overlapping patches (conv has no bias in CLIP), fp32 result
        h, B, S, hn0 = self._stem(x, want_hn0=True)
        h = self.encoder.run(h, B, S, causal=False, hn0=hn0)
        return self._head(h, B, S)

    @torch.jit.unused
    def forward_patches(self, patches: Tensor) -> Tensor:
        """Inference entry for a device-side loader (extension; transforms.clip_transform.CLIPImageTransform.patches): bf16 im2col
        rows [B*G2, Kpad] (column (c*P+py)*P+px, Kpad = 3*P*P rounded up to 64) instead of the fp32 image -- the same rows
        `forward` builds with mmamd_patchify, so the result is identical."""
        self._check_patches(patches)
        h, B, S = self._stem_patches(patches)
        h = self.encoder.run(h, B, S, causal=False)
        return self._head(h, B, S)

    @torch.jit.unused
    def _check_patches(self, patches: Tensor) -> None:
        g = self.image_size // self.patch_size
        G2 = g * g
        K = 3 * self.patch_size * self.patch_size
        kpad = (K + 63) // 64 * 64
        if patches.dim() != 2 or patches.shape[1] != kpad or patches.shape[0] % G2 != 0 or patches.dtype != torch.bfloat16:
            raise ValueError(f"Expected bf16 patch rows [B*{G2}, {kpad}], found {patches.dtype} {tuple(patches.shape)}")

    @torch.jit.unused
    def _stem(self, x: Tensor, want_hn0: bool = False):
        """image (fp32 or bf16) -> (residual stream fp32 [B*S, w], B, S[, hn0]): the part of the inference forward in front of the layer stack.
        16- and 32-pixel patches take the fused stem: the patch-embedding GEMM gathers the patch rows from the bf16 image itself (LDS-DMA source
        addresses: no im2col matrix in HBM) and adds the positional embedding in its epilogue; ONE row kernel then writes the CLS rows, applies
        ln_pre and — `want_hn0` — also norm1 of the first encoder layer (`hn0`, bf16).  Bit-identical to the patchify / GEMM / assemble path
        (tests/test_gpu_fused_stem.py).  14-pixel patches (28-byte runs, not DMA-able) keep that path."""
        P = self.patch_size
        K = 3 * P * P
        xc = x if x.is_contiguous() else x.contiguous()
        # (other shapes -- an image side that is not a whole number of patches, where nn.Conv2d(stride = P) floors the grid; a stack without
        #  layers -- keep the patchify / assemble path)
        if (P in (16, 32) and self.image_size % 8 == 0 and self.image_size % P == 0 and self.width % 8 == 0 and xc.numel() * 2 < 2**32
                and (not want_hn0 or len(self.encoder.layers) > 0)):
            f32, bf = torch.float32, torch.bfloat16
            pk = self._packed.get
            B = xc.shape[0]
            g = self.image_size // P
            S = g * g + 1
            img = xc if xc.dtype == bf else ops.convert(xc, bf)  # (the im2col rows were bf16 on the other path too: same rounding)
            pos = pk(self.positional_embedding, f32)
            h = ops.patch_embed_fused(img, self._conv_weight_bf16(K), pos, P)
            ln1 = None
            if want_hn0:
                n1 = self.encoder.layers[0].norm1
                ln1 = (self.encoder._packed.get(n1.weight, f32), self.encoder._packed.get(n1.bias, f32), n1.eps)
            hn0 = ops.vit_cls_lnpre_ln(h, pk(self.cls_token_embedding, f32), pos, pk(self.ln_pre.weight, f32), pk(self.ln_pre.bias, f32),
                                       self.ln_pre.eps, B, S, ln1)
            return (h, B, S, hn0) if want_hn0 else (h, B, S)
        res = self._stem_patches(ops.patchify(xc, P, (K + 63) // 64 * 64))
        return (*res, None) if want_hn0 else res

    @torch.jit.unused
    def _stem_patches(self, patches: Tensor):
        f32 = torch.float32
        pk = self._packed.get
        g = self.image_size // self.patch_size
        G2 = g * g
        B = patches.shape[0] // G2
        pe = ops.gemm_bf16(patches, self._conv_weight_bf16(patches.shape[1]), out_dtype=f32)
        # prepend CLS, + positional embedding, ln_pre  -> fp32 residual stream [B*S, w]
        h = ops.vit_assemble_ln(pe, pk(self.cls_token_embedding, f32), pk(self.positional_embedding, f32),
                                pk(self.ln_pre.weight, f32), pk(self.ln_pre.bias, f32), self.ln_pre.eps, B, G2)
        return h, B, G2 + 1

    @torch.jit.unused
    def _head(self, h: Tensor, B: int, S: int) -> Tensor:
        """K7: ln_post(CLS) @ projection -- the part behind the layer stack."""
        f32 = torch.float32
        pk = self._packed.get
        out = ops.pool_ln_proj(h, B, S, None, pk(self.ln_post.weight, f32), pk(self.ln_post.bias, f32),
                               self.ln_post.eps, pk(self.projection, f32), proj_is_linear_weight=False)
        return out if self.projection.dtype == f32 else ops.convert(out, self.projection.dtype)

    @torch.jit.unused
    def _forward_train(self, x: Tensor) -> Tensor:
        """Differentiable forward (train mode, grad enabled): the autograd nodes of models/clip/_train.py."""
        x0, B, S = self._train_stem(x)
        return self._train_head(_train.run_stack(self.encoder, x0, B, S, False), B, S)

    @torch.jit.unused
    def _train_stem(self, x: Tensor):
        """Differentiable patch embedding + CLS + positions + ln_pre -> (fp32 residual stream [B*S, w], B, S)."""
        B = x.size(0)
        g = self.image_size // self.patch_size
        S = g * g + 1
        x0 = _train.VisionEmbedFn.apply(x, self.conv.weight, self.cls_token_embedding, self.positional_embedding, self.ln_pre.weight,
                                        self.ln_pre.bias, self.patch_size, self.ln_pre.eps)
        return x0, B, S

    @torch.jit.unused
    def _train_head(self, h: Tensor, B: int, S: int) -> Tensor:
        """Differentiable ln_post over the CLS rows + projection."""
        cls_rows = torch.arange(0, B * S, S, dtype=torch.int64, device=h.device)
        return _train.PooledHeadFn.apply(h, cls_rows, self.ln_post.weight, self.ln_post.bias, self.projection, self.ln_post.eps, False)
