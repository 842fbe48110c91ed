"""Host-side mirror of torchmultimodal/models/coca/coca_model.py (MultimodalOutput :25-31, CoCaModel :34-135, coca_vit
:138-376, coca_vit_b_32 / coca_vit_l_14 :379-431, CoCaForPretraining :434-466, coca_for_pretraining :469-471,
CoCaModelWithHeads :477-508).  Same constructors, attribute names, state_dict keys and initialisation order.

MI355X execution: ViT (no CLS row) -> attention pooler(s) (batch-shared learned queries, 64/96-wide heads) -> fp32 projection +
L2 normalise; causal text decoder with the padding-aware CLS mask on a second HIP stream; multimodal decoder with
cross-attention; vocabulary GEMM; captioning cross entropy (ignore_index = pad) and the contrastive loss kernels.
"""

import math
from functools import partial
from typing import Any, Callable, Dict, List, NamedTuple, Optional, Tuple, Union

import torch
from torch import nn, Tensor

from ... import _torch_ops, ops
from ..._packing import PackedCache, PackedModeMixin
from ...modules.encoders.vision_transformer import vision_transformer
from ...modules.layers.attention_pooler import AttentionPooler, CascadedAttentionPooler
from ...modules.layers.transformer import TransformerOutput
from ...modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
from ...modules.losses.flava import cls_linear
from ...schedule import get_schedule, train_side_stream_now
from ..._autograd import CrossEntropyFn, L2NormalizeFn, wants_grad
from .multimodal_decoder import CoCaMultimodalDecoder
from .text_decoder import CoCaTextDecoder


_torch_ops.try_load()


class MultimodalOutput(NamedTuple):
    image_pooled_output: Tensor
    text_pooled_output: Tensor
    multimodal_embeddings: Tensor
    multimodal_pooled_embeddings: Optional[Tensor] = None


_SIDE_STREAMS = {}


def _side_stream(device: torch.device) -> torch.cuda.Stream:
    s = _SIDE_STREAMS.get(device)
    if s is None:
        s = _SIDE_STREAMS[device] = torch.cuda.Stream(device=device)
    return s


class CoCaModel(PackedModeMixin, nn.Module):
    def __init__(self, vision_encoder: nn.Module, text_decoder: CoCaTextDecoder, multimodal_decoder: CoCaMultimodalDecoder,
                 vision_pooler: nn.Module, vision_proj: nn.Module):
        super().__init__()
        self.vision_encoder = vision_encoder
        self.text_decoder = text_decoder
        self.multimodal_decoder = multimodal_decoder
        self.vision_pooler = vision_pooler
        self.vision_proj = vision_proj
        self._packed = PackedCache()

    def forward(self, images: Tensor, texts: Tensor, text_padding_mask: Optional[Tensor] = None) -> MultimodalOutput:
        if torch.jit.is_scripting():
            return self._forward_ops(images, texts, text_padding_mask)
        else:
            return self._forward_host(images, texts, text_padding_mask)

    def _forward_ops(self, images: Tensor, texts: Tensor, text_padding_mask: Optional[Tensor]) -> MultimodalOutput:
        """The forward through the dispatcher ops (torch.ops.mmamd.*, csrc/torch_ops.cpp) — what torch.jit.script sees (reference:
        tests/models/coca/test_coca_model.py:146-154 scripts the model) and what torch.compile traces.  Inference, one stream; the same
        kernels as the eager forward (the stacked q|k|v projections of the decoders run as three GEMMs)."""
        pooled_text_embeddings, text_tokens = self.text_decoder(texts, text_padding_mask)
        contrastive_text_embeddings = torch.ops.mmamd.l2_normalize(pooled_text_embeddings, 1e-12)
        image_embeddings = self.vision_encoder(images).last_hidden_state
        assert image_embeddings is not None, "Image embeddings must be Tensor"
        pooled_outputs = self.vision_pooler(image_embeddings)
        if isinstance(pooled_outputs, list):
            assert len(pooled_outputs) == 2
            captioning_image_embeddings, contrastive = pooled_outputs[0], pooled_outputs[1]
            B, nq, D = contrastive.size(0), contrastive.size(1), contrastive.size(2)
            proj = torch.ops.mmamd.rows_linear_f32(contrastive.contiguous().view(B * nq, D), self.vision_proj.weight, self.vision_proj.bias)
            contrastive_image_embeddings = torch.ops.mmamd.l2_normalize(proj, 1e-12).view(B, nq, -1)
        else:
            assert isinstance(pooled_outputs, Tensor), "Pooled image embeddings must be Tensor"
            captioning_image_embeddings = pooled_outputs[:, 1:]
            proj = torch.ops.mmamd.rows_linear_f32(pooled_outputs[:, 0].contiguous(), self.vision_proj.weight, self.vision_proj.bias)
            contrastive_image_embeddings = torch.ops.mmamd.l2_normalize(proj, 1e-12)
        multimodal_embeddings = self.multimodal_decoder(text_tokens, captioning_image_embeddings)
        return MultimodalOutput(contrastive_image_embeddings, contrastive_text_embeddings, multimodal_embeddings)

    @torch.jit.unused
    def _forward_host(self, images: Tensor, texts: Tensor, text_padding_mask: Optional[Tensor] = None) -> MultimodalOutput:
        if torch.compiler.is_compiling() and not wants_grad(self):
            return self._forward_ops(images, texts, text_padding_mask)
        training = wants_grad(self, images)
        l2n = L2NormalizeFn.apply if training else ops.l2_normalize
        dev = images.device
        side = None
        if training and not train_side_stream_now():  # differentiable path on one stream
            pooled_text_embeddings, text_tokens = self.text_decoder(texts, text_padding_mask)
            contrastive_text_embeddings = l2n(pooled_text_embeddings)
        elif dev.type == "cuda":  # text decoder on a side stream: its small grids fill the CUs the ViT leaves idle (in training the autograd
            # engine runs the decoder's backward on that stream too: schedule.train_side_stream)
            main = torch.cuda.current_stream(dev)
            side = _side_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                pooled_text_embeddings, text_tokens = self.text_decoder(texts, text_padding_mask)
                contrastive_text_embeddings = l2n(pooled_text_embeddings)
        else:
            pooled_text_embeddings, text_tokens = self.text_decoder(texts, text_padding_mask)  # raises: no CPU path
            contrastive_text_embeddings = ops.l2_normalize(pooled_text_embeddings)

        vision_encoder_outs = self.vision_encoder(images)
        if isinstance(vision_encoder_outs, TransformerOutput):
            image_embeddings = vision_encoder_outs.last_hidden_state
        elif isinstance(vision_encoder_outs, tuple):
            image_embeddings = vision_encoder_outs[0]
        else:
            image_embeddings = vision_encoder_outs
        assert isinstance(image_embeddings, Tensor), "Image embeddings must be Tensor"
        pooled_outputs = self.vision_pooler(image_embeddings)
        if isinstance(pooled_outputs, list):
            assert len(pooled_outputs) == 2
            captioning_image_embeddings, contrastive_image_embeddings = pooled_outputs
            B, nq, D = contrastive_image_embeddings.shape
            # [B, 1, D] stays [B, 1, D], exactly like the reference (its cascaded pooler keeps the query dimension)
            proj = cls_linear(contrastive_image_embeddings.reshape(B * nq, D), self.vision_proj, self._packed)
            contrastive_image_embeddings = l2n(proj).view(B, nq, -1)
        else:
            assert isinstance(pooled_outputs, Tensor), "Pooled image embeddings must be Tensor"
            # contrastive = pooled[:, 0] (projected straight out of the pooled tensor), captioning = pooled[:, 1:]
            captioning_image_embeddings = pooled_outputs[:, 1:]
            contrastive_image_embeddings = l2n(cls_linear(pooled_outputs, self.vision_proj, self._packed))

        if side is not None:
            main.wait_stream(side)
            for t in (pooled_text_embeddings, text_tokens, contrastive_text_embeddings):
                t.record_stream(main)
        multimodal_embeddings = self.multimodal_decoder(text_tokens, captioning_image_embeddings)
        return MultimodalOutput(contrastive_image_embeddings, contrastive_text_embeddings, multimodal_embeddings)


def coca_vit(
    *,
    vision_patch_size: int,
    vision_dim_feedforward: int,
    vision_n_layer: int,
    vision_n_head: int,
    vocab_size: int,
    num_text_positions: int,
    text_hidden_dim: int,
    text_n_layer: int,
    text_n_head: int,
    text_dim_feedforward: int,
    text_output_dim: int,
    fusion_n_layer: int,
    fusion_n_head: int,
    fusion_dim_feedforward: int,
    pooler_input_embed_dim: int,
    pooler_output_embed_dim: int,
    pooler_n_head: int,
    image_size: Union[int, Tuple[int, int]] = 224,
    num_channels: int = 3,
    vision_activation: Callable[..., nn.Module] = nn.GELU,
    vision_transformer_dropout: float = 0.0,
    patch_embed_dropout_prob: float = 0.0,
    vision_layer_norm_eps: float = 1e-5,
    vision_final_layer_norm_eps: Optional[float] = None,
    vision_norm_first: bool = True,
    vision_include_cls_embed: bool = False,  # This is different from ViT default
    vision_drop_path_rate: Optional[float] = None,
    vision_patch_drop_rate: Optional[Union[float, Tuple[float, float]]] = None,
    pad_idx: Optional[int] = 0,
    text_embed_cls: bool = True,
    text_dropout: float = 0.0,
    text_activation: Callable[..., nn.Module] = nn.GELU,
    text_layer_norm_eps: float = 1e-5,
    text_norm_first: bool = True,
    text_final_layer_norm_eps: Optional[float] = 1e-5,
    fusion_dropout: float = 0.0,
    fusion_activation: Callable[..., nn.Module] = nn.GELU,
    fusion_layer_norm_eps: float = 1e-5,
    fusion_norm_first: bool = True,
    fusion_final_layer_norm_eps: Optional[float] = 1e-5,
    multimodal_output_projection_dim: Optional[int] = None,
    cascaded_pooler: bool = True,
    pooler_n_queries: int = 256,
    pooler_layer_norm_eps: float = 1e-5,
) -> CoCaModel:
    attention_pooler: nn.Module
    if cascaded_pooler:
        captioning_pooler = AttentionPooler(input_embed_dim=pooler_input_embed_dim, output_embed_dim=pooler_output_embed_dim,
                                            n_head=pooler_n_head, n_queries=pooler_n_queries, layer_norm_eps=pooler_layer_norm_eps)
        contrastive_pooler = AttentionPooler(input_embed_dim=pooler_output_embed_dim, output_embed_dim=pooler_output_embed_dim,
                                             n_head=pooler_n_head, n_queries=1, layer_norm_eps=pooler_layer_norm_eps)
        attention_pooler = CascadedAttentionPooler([captioning_pooler, contrastive_pooler])
    else:
        attention_pooler = AttentionPooler(input_embed_dim=pooler_input_embed_dim, output_embed_dim=pooler_output_embed_dim,
                                           n_head=pooler_n_head, n_queries=pooler_n_queries + 1, layer_norm_eps=pooler_layer_norm_eps)
    vision_proj = nn.Linear(pooler_output_embed_dim, pooler_output_embed_dim, bias=False)
    nn.init.normal_(vision_proj.weight, std=pooler_input_embed_dim**-0.5)
    vision_encoder = vision_transformer(
        patch_size=vision_patch_size, hidden_dim=pooler_input_embed_dim, dim_feedforward=vision_dim_feedforward,
        n_layer=vision_n_layer, n_head=vision_n_head, image_size=image_size, num_channels=num_channels,
        activation=vision_activation, transformer_dropout=vision_transformer_dropout,
        patch_embed_dropout_prob=patch_embed_dropout_prob, layer_norm_eps=vision_layer_norm_eps,
        final_layer_norm_eps=vision_final_layer_norm_eps, norm_first=vision_norm_first,
        include_cls_embed=vision_include_cls_embed, drop_path_rate=vision_drop_path_rate, patch_drop_rate=vision_patch_drop_rate)
    text_decoder = CoCaTextDecoder(
        vocab_size=vocab_size, num_positions=num_text_positions, embedding_dim=text_hidden_dim, n_layer=text_n_layer,
        n_head=text_n_head, dim_feedforward=text_dim_feedforward, output_dim=text_output_dim, pad_idx=pad_idx,
        embed_cls=text_embed_cls, dropout=text_dropout, activation=text_activation, layer_norm_eps=text_layer_norm_eps,
        norm_first=text_norm_first, final_layer_norm_eps=text_final_layer_norm_eps)
    mm_input_seq_len = num_text_positions - 1 if text_embed_cls else num_text_positions
    multimodal_decoder = CoCaMultimodalDecoder(
        input_seq_len=mm_input_seq_len, text_embedding_dim=pooler_output_embed_dim, n_layer=fusion_n_layer, n_head=fusion_n_head,
        dim_feedforward=fusion_dim_feedforward, output_dim=multimodal_output_projection_dim, dropout=fusion_dropout,
        activation=fusion_activation, layer_norm_eps=fusion_layer_norm_eps, norm_first=fusion_norm_first,
        final_layer_norm_eps=fusion_final_layer_norm_eps)
    return CoCaModel(vision_encoder=vision_encoder, text_decoder=text_decoder, multimodal_decoder=multimodal_decoder,
                     vision_proj=vision_proj, vision_pooler=attention_pooler)


def coca_vit_b_32() -> CoCaModel:
    return coca_vit(
        vision_patch_size=32, vision_n_layer=12, vision_n_head=12, vision_dim_feedforward=3072, vision_include_cls_embed=False,
        vocab_size=49408, num_text_positions=77, text_hidden_dim=512, text_n_layer=12, text_n_head=8, text_dim_feedforward=2048,
        text_output_dim=512, fusion_n_layer=12, fusion_n_head=8, fusion_dim_feedforward=2048,
        multimodal_output_projection_dim=49408, pooler_input_embed_dim=768, pooler_output_embed_dim=512, pooler_n_head=8,
        cascaded_pooler=True)


def coca_vit_l_14() -> CoCaModel:
    return coca_vit(
        vision_patch_size=14, vision_n_layer=24, vision_n_head=16, vision_dim_feedforward=4096, vision_include_cls_embed=False,
        vocab_size=49408, num_text_positions=77, text_hidden_dim=768, text_n_layer=12, text_n_head=12, text_dim_feedforward=3072,
        text_output_dim=768, fusion_n_layer=12, fusion_n_head=12, fusion_dim_feedforward=3072,
        multimodal_output_projection_dim=49408, pooler_input_embed_dim=1024, pooler_output_embed_dim=768, pooler_n_head=8,
        cascaded_pooler=True)


class CoCaForPretraining(PackedModeMixin, nn.Module):
    """CoCa model tied to the captioning and contrastive losses (reference :434-466)."""

    def __init__(self, model: CoCaModel, pad_idx: int = 0, contrastive_logit_scale_min: Optional[float] = math.log(1.0),
                 contrastive_logit_scale_max: Optional[float] = math.log(100.0)):
        super().__init__()
        self.model = model
        self.contrastive_loss = ContrastiveLossWithTemperature(logit_scale_min=contrastive_logit_scale_min,
                                                               logit_scale_max=contrastive_logit_scale_max)
        self.caption_loss = nn.CrossEntropyLoss(ignore_index=pad_idx)

    def forward(self, images: Tensor, texts: Tensor, text_padding_mask: Optional[Tensor] = None) -> Dict[str, Tensor]:
        model_outs = self.model(images, texts, text_padding_mask)
        captioning_labels = texts[:, 1:].contiguous()  # next-token labels: an int64 strided copy (index bookkeeping)
        img = model_outs.image_pooled_output
        if img.dim() != 2:
            raise ops.MmamdError(
                f"image_pooled_output has shape {tuple(img.shape)}: the cascaded pooler keeps the query dimension and the "
                "reference's CoCaForPretraining fails on it too (use coca_vit(..., cascaded_pooler=False))")
        contrastive_loss = self.contrastive_loss(img, model_outs.text_pooled_output)
        mm = model_outs.multimodal_embeddings
        vocab_size = mm.shape[-1]
        logits = mm.flatten(0, 1)  # a view: rows keep the (padded) pitch of the vocabulary GEMM
        if torch.is_grad_enabled() and logits.requires_grad:
            captioning_loss = CrossEntropyFn.apply(logits, captioning_labels.view(-1), self.caption_loss.ignore_index)
        else:
            captioning_loss = ops.cross_entropy(logits, captioning_labels.view(-1), self.caption_loss.ignore_index)
        return {"contrastive": contrastive_loss, "captioning": captioning_loss}


def coca_for_pretraining(pad_idx: int = 0, **kwargs: Any) -> CoCaForPretraining:
    model = coca_vit(**kwargs)
    return CoCaForPretraining(model, pad_idx=pad_idx)


default_coca_cls_pooler = partial(torch.select, dim=1, index=-1)


class CoCaModelWithHeads(PackedModeMixin, nn.Module):
    """CoCa with task heads on the pooled multimodal embeddings (reference :477-508).  The pooler / heads are user modules:
    they run as given (they are not part of the contrastive hot path)."""

    def __init__(self, model: CoCaModel, heads: nn.ModuleDict, pad_idx: int = 0, pooler: Callable = default_coca_cls_pooler):
        super().__init__()
        self.model = model
        self.heads = heads
        self.pooler = pooler

    def forward(self, images: Tensor, texts: Tensor, text_padding_mask: Optional[Tensor] = None) -> Dict[str, Tensor]:
        model_out = self.model(images, texts, text_padding_mask)
        mm_out = model_out.multimodal_embeddings
        bsz = mm_out.shape[0]
        pooled_output = self.pooler(mm_out).view((bsz, -1))
        head_outputs = {}
        for k, head in self.heads.items():
            head_outputs[k] = head(pooled_output)
        return head_outputs
