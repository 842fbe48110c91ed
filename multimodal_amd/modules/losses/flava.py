"""Host-side mirror of the contrastive part of torchmultimodal/modules/losses/flava.py: Pooler (:84-97) and
FLAVAGlobalContrastiveLoss (:241-293) with its output record (:43-52).

The MLM / MIM / ITM heads and FLAVAPretrainingLoss of that file are outside the dual-encoder contrastive path
(out of scope) and are not provided.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, OrderedDict, Union

import torch
from torch import nn, Tensor

from ... import ops
from ..._packing import PackedCache
from ...utils.distributed import BackpropType
from .contrastive_loss_with_temperature import contrastive_loss_with_temperature


@dataclass
class FLAVAGlobalContrastiveLossOutput(OrderedDict):
    text_embedding: Tensor
    image_embedding: Tensor
    logit_scale: Tensor
    image_logits: Tensor
    text_logits: Tensor
    image_loss: Tensor
    text_loss: Tensor
    loss: Tensor


class Pooler(nn.Module):
    """tanh(dense(hidden_states[:, 0])) — one exact-fp32 MFMA kernel that reads the CLS rows in place (row stride S*d),
    adds the bias and applies tanh (csrc/rowops.hip: rows_linear_f32_kernel)."""

    def __init__(self, hidden_size: int = 768, **kwargs: Any):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)
        self.activation = nn.Tanh()
        self._packed = PackedCache()

    def forward(self, hidden_states: Tensor) -> Tensor:
        return cls_linear(hidden_states, self.dense, self._packed, tanh=True)


def cls_linear(hidden_states: Tensor, dense: nn.Linear, packed: PackedCache, tanh: bool = False) -> Tensor:
    """dense(hidden_states[:, 0]) for a contiguous fp32 [B, S, d] tensor (or a [B, d] one), without materialising the slice."""
    if hidden_states.dtype != torch.float32:
        raise ops.MmamdError("pooled projections on the MI355X path take fp32 hidden states")
    if hidden_states.dim() == 3:
        base = hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous()
        B, S, d = base.shape
        stride = S * d
    elif hidden_states.dim() == 2:
        base = hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous()  # e.g. x[:, 0, :]: B*d floats
        (B, d), stride = base.shape, base.shape[1]
    else:
        raise ops.MmamdError("expected [B, S, d] or [B, d] hidden states")
    f32 = torch.float32
    w = packed.get(dense.weight, f32)
    b = packed.get(dense.bias, f32) if dense.bias is not None else None
    return ops.rows_linear_f32(base, stride, B, w, b, tanh=tanh)


class FLAVAGlobalContrastiveLoss(nn.Module):
    def __init__(
        self,
        logit_scale: Union[float, nn.Parameter] = None,
        image_embedding_size: int = 768,
        text_embedding_size: int = 768,
        projection_size: int = 768,
        image_embedding_index: int = 0,
        text_embedding_index: int = 0,
    ):
        super().__init__()
        if logit_scale is None:
            logit_scale = math.log(1 / 0.07)
        # If already initialized, set to what was passed
        if isinstance(logit_scale, nn.Parameter):
            self.logit_scale = logit_scale
        else:
            self.logit_scale = nn.Parameter(logit_scale * torch.ones([]))

    def forward(self, image_sequence: Tensor, text_sequence: Tensor, mask: Tensor) -> FLAVAGlobalContrastiveLossOutput:
        if image_sequence.dim() != 2 or text_sequence.dim() != 2:
            raise ops.MmamdError("FLAVAGlobalContrastiveLoss on the MI355X path takes the projected [B, E] embeddings")
        text_embedding = ops.l2_normalize(_f32c(text_sequence))
        image_embedding = ops.l2_normalize(_f32c(image_sequence))
        ops.clamp_scalar_(self.logit_scale.data.view(1), 0.0, 4.6052)  # reference :276
        output = contrastive_loss_with_temperature(
            embeddings_a=image_embedding,
            embeddings_b=text_embedding,
            logit_scale=self.logit_scale,
            mask=mask,
            backprop_type=BackpropType.GLOBAL,  # always true for the FLAVA global contrastive loss
        )
        return FLAVAGlobalContrastiveLossOutput(
            loss=output.loss,
            image_logits=output.logits_a,
            text_logits=output.logits_b,
            image_loss=output.loss_a,
            text_loss=output.loss_b,
            text_embedding=text_embedding,
            image_embedding=image_embedding,
            logit_scale=self.logit_scale.data,
        )


def _f32c(t: Tensor) -> Tensor:
    t = t.detach()
    t = t if t.is_contiguous() else t.contiguous()
    return t if t.dtype == torch.float32 else ops.convert(t, torch.float32)
