"""Host-side mirror of torchmultimodal/modules/losses/flava.py: Pooler (:84-97), TwoWayHead / ITMLoss (:100-140),
MaskedPredictionHead / MaskedPredictionLoss (:143-238), FLAVAGlobalContrastiveLoss (:241-293), FLAVAPretrainingLoss
(:296-484) and their output records (:30-81).  Same constructors, attribute names and state_dict keys.

MI355X execution of a masked-prediction head: the labelled positions are compacted on the device
(csrc/loss.hip: select_tokens_kernel — the reference's boolean indexing), their rows gathered straight out of the
[B, S, d] sequence as bf16, then dense GEMM (+bias, erf-GELU epilogue) -> LayerNorm -> vocabulary GEMM (+tied bias,
fp32 logits) -> row cross entropy.  Nothing is computed for unlabelled positions.
"""
from __future__ import annotations

import math
import warnings
from dataclasses import dataclass, field
from typing import Any, Callable, Optional, OrderedDict, Tuple, Union

import torch
from torch import nn, Tensor

from ... import ops
from ..._packing import PackedCache
from ...utils.common import ModelOutput
from ...utils.distributed import BackpropType
from ..layers.normalizations import Fp32LayerNorm
from .contrastive_loss_with_temperature import contrastive_loss_with_temperature


def assert_labels_are_present(labels: Optional[Tensor], category: str = "labels") -> None:
    assert labels is not None, f"Model is in training model but {category} are not passed"


@dataclass
class ITMLossOutput(ModelOutput):
    logits: Tensor
    loss: Tensor


@dataclass
class MaskedPredictionLossOutput(ModelOutput):
    logits: Tensor
    loss: Tensor


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


@dataclass
class FLAVAPretrainingLossesCollection(ModelOutput):
    mmm_text_loss: Optional[Tensor] = None
    mmm_image_loss: Optional[Tensor] = None
    mim_loss: Optional[Tensor] = None
    mlm_loss: Optional[Tensor] = None
    itm_loss: Optional[Tensor] = None
    global_contrastive_loss: Optional[Tensor] = None


@dataclass
class FLAVAPretrainingLossOutput(ModelOutput):
    losses: FLAVAPretrainingLossesCollection = field(default_factory=FLAVAPretrainingLossesCollection)
    mlm_output: Optional[MaskedPredictionLossOutput] = None
    mim_output: Optional[MaskedPredictionLossOutput] = None
    mmm_text_output: Optional[MaskedPredictionLossOutput] = None
    mmm_image_output: Optional[MaskedPredictionLossOutput] = None
    itm_output: Optional[ITMLossOutput] = None
    global_contrastive_output: Optional[FLAVAGlobalContrastiveLossOutput] = None
    image_sequence: Optional[Tensor] = None
    text_sequence: Optional[Tensor] = None
    image_masked_sequence: Optional[Tensor] = None
    text_masked_sequence: Optional[Tensor] = None
    multimodal_sequence: Optional[Tensor] = None
    multimodal_masked_sequence: Optional[Tensor] = None


class Pooler(nn.Module):
    """tanh(dense(hidden_states[:, 0])) — one exact-fp32 MFMA kernel that reads the CLS rows in place (row stride S*d),
    adds the bias and applies tanh (csrc/rowops.hip: rows_linear_f32_kernel)."""

    def __init__(self, hidden_size: int = 768, **kwargs: Any):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)
        self.activation = nn.Tanh()
        self._packed = PackedCache()

    def forward(self, hidden_states: Tensor) -> Tensor:
        if torch.is_grad_enabled() and hidden_states.requires_grad and hidden_states.dim() == 3:
            from ..._autograd import TanhRowsLinearFn  # differentiable pooler (ITM head)

            B, S, d = hidden_states.shape
            rows = torch.arange(0, B * S, S, dtype=torch.int64, device=hidden_states.device)
            return TanhRowsLinearFn.apply(hidden_states.reshape(B * S, d), rows, self.dense.weight, self.dense.bias)
        return cls_linear(hidden_states.detach(), self.dense, self._packed, tanh=True)


def cls_linear(hidden_states: Tensor, dense: nn.Linear, packed: PackedCache, tanh: bool = False) -> Tensor:
    """dense(hidden_states[:, 0]) for a contiguous fp32 [B, S, d] tensor (or a [B, d] one), without materialising the slice."""
    if hidden_states.dtype != torch.float32:
        raise ops.MmamdError("pooled projections on the MI355X path take fp32 hidden states")
    if not tanh and torch.is_grad_enabled() and (hidden_states.requires_grad or dense.weight.requires_grad and dense.training):
        from ..._autograd import RowsLinearFn  # differentiable path: gather the first row of every sample, Linear, both on HIP

        if hidden_states.dim() == 3:
            B, S, d = hidden_states.shape
            x2d = hidden_states.reshape(B * S, d)
            rows = torch.arange(0, B * S, S, dtype=torch.int64, device=hidden_states.device)
        else:
            x2d = hidden_states
            rows = torch.arange(hidden_states.shape[0], dtype=torch.int64, device=hidden_states.device)
        return RowsLinearFn.apply(x2d, rows, dense.weight, dense.bias)
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
        if torch.is_grad_enabled() and (image_sequence.requires_grad or text_sequence.requires_grad):
            from ..._autograd import L2NormalizeFn  # differentiable: normalise and loss nodes with HIP forward + backward

            text_embedding = L2NormalizeFn.apply(text_sequence)
            image_embedding = L2NormalizeFn.apply(image_sequence)
        else:
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


class TwoWayHead(nn.Module):
    def __init__(self, hidden_size: int = 768, **kwargs: Any):
        super().__init__()
        self.seq_relationship = nn.Linear(hidden_size, 2)
        self._packed = PackedCache()

    def forward(self, pooled_output: Tensor) -> Tensor:
        return cls_linear(pooled_output, self.seq_relationship, self._packed)


class ITMLoss(nn.Module):
    def __init__(self, hidden_size: int = 768, ignore_index: int = -1, **kwargs: Any):
        super().__init__()
        self.pooler = Pooler(hidden_size=hidden_size)
        self.cls = TwoWayHead(hidden_size=hidden_size)
        self.ce_loss = nn.CrossEntropyLoss(ignore_index=ignore_index)

    def forward(self, hidden_states: Tensor, labels: Tensor) -> ITMLossOutput:
        if self.training:
            assert_labels_are_present(labels, "itm labels")
        pooled_output = self.pooler(hidden_states)
        scores = self.cls(pooled_output)
        if labels is None:
            loss = torch.zeros((), dtype=torch.float32, device=scores.device)
        else:
            lab = labels.reshape(-1)
            lab = lab if lab.is_contiguous() else lab.contiguous()
            if torch.is_grad_enabled() and scores.requires_grad:
                from ..._autograd import CrossEntropyFn

                loss = CrossEntropyFn.apply(scores.view(-1, 2), lab, self.ce_loss.ignore_index)
            else:
                loss = ops.cross_entropy(scores.view(-1, 2), lab, self.ce_loss.ignore_index)
        return ITMLossOutput(logits=scores, loss=loss)


class MaskedPredictionHead(nn.Module):
    def __init__(self, hidden_size: int = 768, vocab_size: int = 30522,
                 transform_act_fn: Callable[[Tensor], Tensor] = nn.functional.gelu, layer_norm_eps: float = 1e-5,
                 use_fp32_layer_norm: bool = True, **kwargs: Any):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)
        self.transform_act_fn = transform_act_fn
        self.layer_norm: nn.LayerNorm
        if use_fp32_layer_norm:
            self.layer_norm = Fp32LayerNorm(hidden_size, eps=layer_norm_eps)
        else:
            self.layer_norm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
        # The output weights are the same as the input embeddings, but there is an output-only bias for each token.
        self.decoder = nn.Linear(hidden_size, vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(vocab_size))
        # link between the two variables so that the bias is correctly resized with `resize_token_embeddings`
        self.decoder.bias = self.bias
        self._packed = PackedCache()

    def run(self, rows: Tensor) -> Tensor:
        """rows: bf16 [N, d] -> fp32 logits [N, vocab] (a view of a buffer whose rows are padded to a multiple of 8)."""
        if self.transform_act_fn is not nn.functional.gelu:
            raise ops.MmamdError("MaskedPredictionHead on the MI355X path: transform_act_fn must be nn.functional.gelu "
                                 "(the only fused epilogue of this head)")
        pk, bf, f32 = self._packed.get, torch.bfloat16, torch.float32
        if rows.shape[0] == 0:  # no labelled position: empty logits (and a NaN mean loss downstream), like the reference
            return torch.empty((0, self.decoder.out_features), dtype=f32, device=rows.device)
        h = ops.gemm_bf16(rows, pk(self.dense.weight, bf), pk(self.dense.bias, f32), act=ops.ACT_GELU_ERF, out_dtype=f32)
        h = ops.layernorm(h, pk(self.layer_norm.weight, f32), pk(self.layer_norm.bias, f32), self.layer_norm.eps, out_dtype=bf)
        V = self.decoder.out_features
        w = self._packed.get_padded_rows(self.decoder.weight, bf, 8)
        b = self._packed.get_padded_rows(self.bias, f32, 8)
        logits = ops.gemm_bf16(h, w, b, out_dtype=f32)
        return logits if logits.shape[1] == V else logits[:, :V]

    def forward(self, hidden_states: Tensor) -> Tensor:
        x = hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous()
        d = x.shape[-1]
        y = self.run(ops.convert(x.view(-1, d), torch.bfloat16))
        return y.unflatten(0, x.shape[:-1])  # keeps the padded row stride: no copy of the logits


class MaskedPredictionLoss(nn.Module):
    def __init__(self, hidden_size: int = 768, vocab_size: int = 30522,
                 transform_act_fn: Callable[[Tensor], Tensor] = nn.functional.gelu, layer_norm_eps: float = 1e-5,
                 ignore_index: int = -1, ignore_nan: bool = False, **kwargs: Any):
        super().__init__()
        self.cls = MaskedPredictionHead(hidden_size=hidden_size, vocab_size=vocab_size, transform_act_fn=transform_act_fn,
                                        layer_norm_eps=layer_norm_eps)
        self.ignore_index = ignore_index
        self.vocab_size = vocab_size
        self.ce_loss = nn.CrossEntropyLoss(ignore_index=ignore_index)
        self.ignore_nan = ignore_nan

    def run(self, base: Tensor, tok_offset: int, L: int, masked_labels: Optional[Tensor],
            row_keep: Optional[Tensor] = None) -> MaskedPredictionLossOutput:
        """Loss over base[:, tok_offset:tok_offset+L, :] (base: contiguous fp32 [B, S, d]) without slicing it: only the
        labelled positions of the samples kept by `row_keep` (uint8 [B]) are gathered and pushed through the head."""
        if base.dtype != torch.float32 or base.dim() != 3 or (not base.is_contiguous() and not base.requires_grad):
            raise ops.MmamdError("MaskedPredictionLoss on the MI355X path takes contiguous fp32 [B, S, d] sequences")
        B, S, d = base.shape
        if masked_labels is None:
            # reference :216-222: every position goes through the head and the loss is 0
            if row_keep is not None:
                raise ops.MmamdError("row filtering without labels is not implemented on the MI355X path")
            lab = torch.zeros((B, L), dtype=torch.int64, device=base.device)
            idx, _ = ops.select_tokens(lab, self.ignore_index if self.ignore_index != 0 else 1, S, tok_offset)
            logits = self.cls.run(ops.gather_rows(base, d, idx, d, torch.bfloat16))
            return MaskedPredictionLossOutput(logits=logits.unflatten(0, (B, L)),
                                              loss=torch.zeros((), dtype=torch.float32, device=base.device))
        if masked_labels.shape[-1] != L or masked_labels.shape[0] != B:
            raise ops.MmamdError(f"masked_labels shape {tuple(masked_labels.shape)} does not match the {B} x {L} positions")
        lab2d = masked_labels.reshape(B, L)
        idx, lab = ops.select_tokens(lab2d if lab2d.is_contiguous() else lab2d.contiguous(), self.ignore_index, S, tok_offset,
                                     row_keep)
        if torch.is_grad_enabled() and (base.requires_grad or (self.training and self.cls.dense.weight.requires_grad)) and idx.numel() > 0:
            from ..._autograd import MaskedHeadLossFn  # differentiable head: forward and backward on the HIP kernels

            if self.cls.transform_act_fn is not nn.functional.gelu:
                raise ops.MmamdError("MaskedPredictionHead on the MI355X path: transform_act_fn must be nn.functional.gelu")
            c = self.cls
            masked_loss, prediction = MaskedHeadLossFn.apply(base, idx, lab, c.dense.weight, c.dense.bias, c.layer_norm.weight,
                                                             c.layer_norm.bias, c.decoder.weight, c.bias, c.layer_norm.eps, self.ignore_index)
            return MaskedPredictionLossOutput(logits=prediction, loss=masked_loss)
        prediction = self.cls.run(ops.gather_rows(base.detach(), d, idx, d, torch.bfloat16))
        masked_loss = ops.cross_entropy(prediction, lab, self.ignore_index)
        if self.ignore_nan and idx.numel() == 0:  # the only way this mean is NaN: no labelled position (:232-235)
            warnings.warn("NaN detected in masked_loss. Replacing it with 0.")
            masked_loss = torch.zeros((), dtype=torch.float32, device=base.device)
        return MaskedPredictionLossOutput(logits=prediction, loss=masked_loss)

    def forward(self, hidden_states: Tensor, masked_labels: Optional[Tensor] = None) -> MaskedPredictionLossOutput:
        if self.training:
            assert_labels_are_present(masked_labels, "masked labels")
        x = hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous()
        if x.dim() != 3:
            raise ops.MmamdError("MaskedPredictionLoss on the MI355X path takes [B, L, d] hidden states")
        return self.run(x, 0, x.shape[1], masked_labels)


def _base3(t: Tensor) -> Tensor:
    if not (torch.is_grad_enabled() and t.requires_grad):
        t = t.detach()
    if t.dtype != torch.float32:
        raise ops.MmamdError("FLAVAPretrainingLoss on the MI355X path takes fp32 sequences")
    return t if t.is_contiguous() else t.contiguous()


def _scaled(loss: Tensor, w: float) -> Tensor:
    return loss if w == 1.0 else loss * w  # scalar loss weighting: host-level glue on a 0-dim tensor


class FLAVAPretrainingLoss(nn.Module):
    def __init__(
        self,
        logit_scale: Union[float, nn.Parameter] = None,
        hidden_size: int = 768,
        text_vocab_size: int = 30522,
        image_vocab_size: int = 8192,
        transform_act_fn: Callable[[Tensor], Tensor] = nn.functional.gelu,
        layer_norm_eps: float = 1e-5,
        ignore_index: int = -1,
        mlm_weight: float = 1.0,
        mim_weight: float = 1.0,
        contrastive_loss_weight: float = 1.0,
        mmm_image_loss_weight: float = 1.0,
        mmm_text_loss_weight: float = 1.0,
        itm_loss_weight: float = 1.0,
        **kwargs: Any,
    ):
        super().__init__()
        self.contrastive_loss = FLAVAGlobalContrastiveLoss(logit_scale=logit_scale, image_embedding_size=hidden_size,
                                                           text_embedding_size=hidden_size, projection_size=hidden_size)
        mk = lambda v: MaskedPredictionLoss(hidden_size=hidden_size, vocab_size=v, transform_act_fn=transform_act_fn,
                                            layer_norm_eps=layer_norm_eps, ignore_index=ignore_index)
        self.mlm_loss = mk(text_vocab_size)
        self.mim_loss = mk(image_vocab_size)
        # Create separate weights for MMM loss
        self.mmm_loss = nn.ModuleDict({"mlm": mk(text_vocab_size), "mim": mk(image_vocab_size)})
        self.itm_loss = ITMLoss(hidden_size=hidden_size, ignore_index=ignore_index)
        self.mim_weight = mim_weight
        self.mlm_weight = mlm_weight
        self.contrastive_loss_weight = contrastive_loss_weight
        self.mmm_image_loss_weight = mmm_image_loss_weight
        self.mmm_text_loss_weight = mmm_text_loss_weight
        self.itm_loss_weight = itm_loss_weight

    def forward(
        self,
        image_sequence: Optional[Tensor] = None,
        text_sequence: Optional[Tensor] = None,
        image_masked_sequence: Optional[Tensor] = None,
        text_masked_sequence: Optional[Tensor] = None,
        multimodal_sequence: Optional[Tensor] = None,
        multimodal_masked_sequence: Optional[Tensor] = None,
        itm_labels: Optional[Tensor] = None,
        mim_labels: Optional[Tensor] = None,
        mlm_labels: Optional[Tensor] = None,
        projected_image_embeddings: Optional[Tensor] = None,
        projected_text_embeddings: Optional[Tensor] = None,
    ) -> FLAVAPretrainingLossOutput:
        outputs = FLAVAPretrainingLossOutput()
        pos_mask = None
        row_keep = None
        # unimodal MIM / MLM: only when there is no multimodal sequence (reference :391-416)
        if image_masked_sequence is not None and self.mim_weight > 0 and multimodal_masked_sequence is None:
            seq = _base3(image_masked_sequence)
            L = mim_labels.size(1) if mim_labels is not None else seq.shape[1] - 1  # CLS row removed
            outputs.mim_output = self.mim_loss.run(seq, seq.shape[1] - L, L, mim_labels)
            outputs.mim_output.loss = _scaled(outputs.mim_output.loss, self.mim_weight)
            outputs.losses.mim_loss = outputs.mim_output.loss
        if text_masked_sequence is not None and self.mlm_weight > 0 and multimodal_masked_sequence is None:
            seq = _base3(text_masked_sequence)
            L = mlm_labels.size(1) if mlm_labels is not None else seq.shape[1] - 1
            outputs.mlm_output = self.mlm_loss.run(seq, seq.shape[1] - L, L, mlm_labels)
            outputs.mlm_output.loss = _scaled(outputs.mlm_output.loss, self.mlm_weight)
            outputs.losses.mlm_loss = outputs.mlm_output.loss

        mm = _base3(multimodal_masked_sequence) if multimodal_masked_sequence is not None else None
        if mm is not None and self.itm_loss_weight > 0:
            B = mm.shape[0]
            if itm_labels is not None:
                # pos_mask = itm_labels != 0, or all-True when no pair is positive (:419-423)
                flags = ops.key_mask(itm_labels.reshape(-1).contiguous())
                kept, _ = ops.select_tokens(itm_labels.reshape(-1).contiguous(), 0, 1, 0)
                if kept.numel() == 0:
                    flags = None
            else:
                flags = None
            pos_mask = flags.view(torch.bool) if flags is not None else torch.ones(B, dtype=torch.bool, device=mm.device)
            row_keep = flags
            outputs.itm_output = self.itm_loss(mm, itm_labels)
            outputs.itm_output.loss = _scaled(outputs.itm_output.loss, self.itm_loss_weight)
            outputs.losses.itm_loss = outputs.itm_output.loss

        if mm is not None and self.mmm_text_loss_weight > 0:
            L = mlm_labels.size(1) if mlm_labels is not None else text_masked_sequence.size(1) - 1
            outputs.mmm_text_output = self.mmm_loss.mlm.run(mm, mm.shape[1] - L, L, mlm_labels, row_keep)
            outputs.mmm_text_output.loss = _scaled(outputs.mmm_text_output.loss, self.mmm_text_loss_weight)
            outputs.losses.mmm_text_loss = outputs.mmm_text_output.loss

        if mm is not None and self.mmm_image_loss_weight > 0:
            # starts from 2 because of 2 CLS rows: the multimodal encoder's and the image encoder's (:455-456)
            total = mim_labels.size(1) if mlm_labels is not None else image_masked_sequence.size(1) - 1
            outputs.mmm_image_output = self.mmm_loss.mim.run(mm, 2, total, mim_labels, row_keep)
            outputs.mmm_image_output.loss = _scaled(outputs.mmm_image_output.loss, self.mmm_image_loss_weight)
            outputs.losses.mmm_image_loss = outputs.mmm_image_output.loss

        if projected_image_embeddings is not None and projected_text_embeddings is not None and self.contrastive_loss_weight > 0:
            outputs.global_contrastive_output = self.contrastive_loss(projected_image_embeddings, projected_text_embeddings, pos_mask)
            outputs.global_contrastive_output.loss = _scaled(outputs.global_contrastive_output.loss, self.contrastive_loss_weight)
            outputs.losses.global_contrastive_loss = outputs.global_contrastive_output.loss

        return outputs
