"""Host-side mirror of torchmultimodal/modules/encoders/vision_transformer.py:19-263 (VisionTransformer, GlobalAveragePooler,
vision_transformer and the vit_* factories) — CoCa's image tower."""

from typing import Any, Callable, Optional, Tuple, Union

import torch
from torch import nn, Tensor

from ... import ops
from ..._autograd import forbid_detached_forward
from ..._packing import PackedCache
from ...utils.common import load_module_from_url
from ..layers.patch_embedding import PatchEmbeddings
from ..layers.transformer import TransformerEncoder, TransformerOutput


class VisionTransformer(nn.Module):
    def __init__(self, embeddings: nn.Module, encoder: nn.Module, pooler: Optional[nn.Module] = None,
                 weight_init_fn: Optional[Callable] = None) -> None:
        super().__init__()
        self.embeddings = embeddings
        self.encoder = encoder
        self.pooler = pooler
        if weight_init_fn:
            self.apply(weight_init_fn)

    def forward(self, images: Tensor, image_patches_mask: Optional[Tensor] = None, attention_mask: Optional[Tensor] = None
                ) -> TransformerOutput:
        embedding_output = self.embeddings(images, image_patches_mask=image_patches_mask).embeddings
        encoder_output = self.encoder(embedding_output, attention_mask=attention_mask, return_hidden_states=True)
        last_hidden_state = encoder_output.last_hidden_state
        if self.pooler is not None:
            assert last_hidden_state is not None, "For pooler, last hidden state cannot be None."
            pooled_output = self.pooler(last_hidden_state)
        else:
            pooled_output = None
        return TransformerOutput(last_hidden_state=last_hidden_state, pooler_output=pooled_output,
                                 hidden_states=encoder_output.hidden_states, attentions=encoder_output.attentions)


class GlobalAveragePooler(nn.Module):
    """Mean over the patch rows (the CLS row is skipped) + LayerNorm + optional Linear head (reference :89-127).  Three kernels: the token
    mean (mmamd_token_mean), the LayerNorm on the B pooled rows, the head in exact fp32 (mmamd_rows_linear_f32).  Inference form."""

    def __init__(self, input_dim: int, output_dim: Optional[int] = None, ln_eps: float = 1e-6,
                 init_weights: Optional[Callable] = None) -> None:
        super().__init__()
        self.norm = nn.LayerNorm(input_dim, eps=ln_eps)
        if output_dim:
            self.head: nn.Module = nn.Linear(input_dim, output_dim)
        else:
            self.head = nn.Identity()
        if init_weights is not None:
            self.apply(init_weights)
        self._packed = PackedCache()

    def forward(self, x: Tensor) -> Tensor:
        forbid_detached_forward(self, x)
        if x.dim() != 3 or x.dtype != torch.float32:
            raise ops.MmamdError("GlobalAveragePooler on the MI355X path takes fp32 [bsz, len, input_dim]")
        if x.shape[1] < 2:
            raise ops.MmamdError("GlobalAveragePooler averages the rows behind the CLS row: the sequence needs at least 2 rows")
        pk, f32 = self._packed.get, torch.float32
        out = ops.token_mean(x if x.is_contiguous() else x.contiguous(), first=1)
        out = ops.layernorm(out, pk(self.norm.weight, f32), pk(self.norm.bias, f32), self.norm.eps, out_dtype=f32)
        if isinstance(self.head, nn.Linear):
            b = pk(self.head.bias, f32) if self.head.bias is not None else None
            out = ops.rows_linear_f32(out, out.shape[1], out.shape[0], pk(self.head.weight, f32), b)
        return out


def vision_transformer(
    *,
    patch_size: int,
    hidden_dim: int,
    dim_feedforward: int,
    n_layer: int,
    n_head: int,
    image_size: Union[int, Tuple[int, int]] = 224,
    num_channels: int = 3,
    activation: Callable[..., nn.Module] = nn.GELU,
    transformer_dropout: float = 0.0,
    patch_embed_dropout_prob: float = 0.0,
    layer_norm_eps: float = 1e-6,
    final_layer_norm_eps: Optional[float] = 1e-6,
    norm_first: bool = True,
    include_cls_embed: bool = True,
    drop_path_rate: Optional[float] = None,
    patch_drop_rate: Optional[Union[float, Tuple[float, float]]] = None,
    pooler: Optional[nn.Module] = None,
    ckpt_path: str = None,
) -> VisionTransformer:
    image_embedding = PatchEmbeddings(image_size=image_size, patch_size=patch_size, hidden_size=hidden_dim,
                                      hidden_dropout_prob=patch_embed_dropout_prob, patch_drop_rate=patch_drop_rate,
                                      num_channels=num_channels, include_cls_embed=include_cls_embed)
    transformer_encoder = TransformerEncoder(n_layer=n_layer, d_model=hidden_dim, n_head=n_head, dim_feedforward=dim_feedforward,
                                             dropout=transformer_dropout, activation=activation, layer_norm_eps=layer_norm_eps,
                                             norm_first=norm_first, final_layer_norm_eps=final_layer_norm_eps,
                                             drop_path_rate=drop_path_rate)
    vit = VisionTransformer(embeddings=image_embedding, encoder=transformer_encoder, pooler=pooler)
    if ckpt_path:
        load_module_from_url(vit, ckpt_path)
    return vit


def vit_b_16(pooler: Optional[nn.Module] = None, **kwargs: Any) -> VisionTransformer:
    return vision_transformer(patch_size=16, n_layer=12, n_head=12, hidden_dim=768, dim_feedforward=3072, pooler=pooler, **kwargs)


def vit_b_32(pooler: Optional[nn.Module] = None, **kwargs: Any) -> VisionTransformer:
    return vision_transformer(patch_size=32, n_layer=12, n_head=12, hidden_dim=768, dim_feedforward=3072, pooler=pooler, **kwargs)


def vit_l_16(pooler: Optional[nn.Module] = None, **kwargs: Any) -> VisionTransformer:
    return vision_transformer(patch_size=16, n_layer=24, n_head=16, hidden_dim=1024, dim_feedforward=4096, pooler=pooler, **kwargs)


def vit_l_32(pooler: Optional[nn.Module] = None, **kwargs: Any) -> VisionTransformer:
    return vision_transformer(patch_size=32, n_layer=24, n_head=16, hidden_dim=1024, dim_feedforward=4096, pooler=pooler, **kwargs)
