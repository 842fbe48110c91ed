"""ContrastiveLossWithTemperature — host-side mirror of
torchmultimodal/modules/losses/contrastive_loss_with_temperature.py:17-201 on the MI355X kernels.

Same signatures, same ContrastiveLossOutput fields, same ValueError, same in-place clamp of the logit_scale
parameter (…:193, done by the clamp_scalar kernel so there is no host sync).  The forward is
    one packed all-gather (RCCL over xGMI; utils/distributed.gather_packed_features)
 -> mmamd_contrastive_fwd (fp32 logits on the exact-f32 MFMA + row cross entropy + reduction).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Dict, Optional, OrderedDict, Union

import torch
from torch import nn, Tensor

from ... import _lib, ops
from ..._custom_op import define as _define
from ...utils.distributed import BackpropType, gather_packed_features


@dataclass
class ContrastiveLossOutput(OrderedDict):
    loss: Tensor
    logits_a: Tensor
    logits_b: Tensor
    loss_a: Tensor
    loss_b: Tensor


_SUPPORTED_CE_KWARGS = {"label_smoothing", "reduction"}


def _as_f32(t: Tensor, keep_row_stride: bool = False) -> Tensor:
    t = t.detach()
    if keep_row_stride and t.dim() == 2 and t.dtype == torch.float32 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t  # a half of CLIP.forward's packed [B, 2E] block: read in place (row stride 2E), gathered without packing copies
    t = t if t.is_contiguous() else t.contiguous()
    if t.dtype == torch.float32:
        return t
    return t.float() if torch.compiler.is_compiling() else ops.convert(t, torch.float32)  # (exact either way)


def contrastive_loss_with_temperature(
    embeddings_a: Tensor,
    embeddings_b: Tensor,
    logit_scale: nn.Parameter,
    mask: Optional[Tensor] = None,
    backprop_type: BackpropType = BackpropType.GLOBAL,
    cross_entropy_kwargs: Optional[Dict[str, Any]] = None,
) -> ContrastiveLossOutput:
    """Functional component for the ContrastiveLossWithTemperature (reference …:50-115).

    Args:
        embeddings_a (Tensor): features from the first modality, [B, E].
        embeddings_b (Tensor): features from the second modality, [B, E].
        logit_scale (nn.Parameter): 0-dim parameter holding the log of the temperature.
        mask (Optional[Tensor]): boolean [B]; rows that are False are dropped from the loss and the logits.
        backprop_type (BackpropType): how gradients flow through the all-gather when the inputs require grad — GLOBAL:
            reduce-scatter of every rank's gathered-feature gradients; LOCAL: this rank's own block only; NONE: none.
        cross_entropy_kwargs: `label_smoothing` and `reduction` ('mean' | 'sum') are supported.
    """
    if not isinstance(backprop_type, BackpropType):
        raise TypeError("backprop_type must be a BackpropType")
    if embeddings_a.shape != embeddings_b.shape or embeddings_a.dim() != 2:
        raise ValueError("embeddings_a and embeddings_b must both be [batch, embedding_dim] of equal shape")
    kwargs = dict(cross_entropy_kwargs or {})
    unknown = set(kwargs) - _SUPPORTED_CE_KWARGS
    if unknown:
        raise NotImplementedError(f"cross_entropy_kwargs {sorted(unknown)} are not supported on the MI355X path")
    smoothing = float(kwargs.get("label_smoothing", 0.0))
    red = kwargs.get("reduction", "mean")
    if red not in ("mean", "sum"):
        raise NotImplementedError(f"cross_entropy reduction '{red}' is not supported on the MI355X path")

    needs_grad = torch.is_grad_enabled() and (embeddings_a.requires_grad or embeddings_b.requires_grad or logit_scale.requires_grad)
    if mask is not None and (mask.dtype != torch.bool or mask.shape != (embeddings_a.shape[0],)):
        raise ValueError("mask must be a boolean tensor of shape (batch,)")
    red_code = _lib.REDUCE_MEAN if red == "mean" else _lib.REDUCE_SUM
    if needs_grad:
        out3, logits_a, logits_b = _ContrastiveFn.apply(embeddings_a, embeddings_b, logit_scale, mask, backprop_type, smoothing, red_code)
    elif torch.compiler.is_compiling():  # the same op as the training forward: nothing but dispatcher ops in the traced graph
        row_mask = mask.contiguous().view(torch.uint8) if mask is not None else None
        out3, logits_a, logits_b, _ = contrastive_fwd_op(_as_f32(embeddings_a), _as_f32(embeddings_b), _scale32(logit_scale), row_mask,
                                                         smoothing, red_code)
    else:
        out3, logits_a, logits_b, _ = _contrastive_forward(embeddings_a, embeddings_b, logit_scale, mask, smoothing, red_code,
                                                           for_backward=False)
    if mask is not None:  # reference …:97-100 returns only the kept rows (data-dependent shape)
        logits_a, logits_b = logits_a[mask], logits_b[mask]
    out_dtype = embeddings_a.dtype
    if out_dtype != torch.float32:
        if torch.compiler.is_compiling():
            logits_a, logits_b, out3 = logits_a.to(out_dtype), logits_b.to(out_dtype), out3.to(out_dtype)
        else:
            logits_a, logits_b = (ops.convert(t.contiguous(), out_dtype) for t in (logits_a, logits_b))
            out3 = out3.to(out_dtype) if needs_grad else ops.convert(out3, out_dtype)
    return ContrastiveLossOutput(loss=out3[0], logits_a=logits_a, logits_b=logits_b, loss_a=out3[1], loss_b=out3[2])


def _scale32(logit_scale: Tensor) -> Tensor:
    scale = logit_scale.detach()
    if scale.dtype != torch.float32:
        scale = scale.float() if torch.compiler.is_compiling() else ops.convert(scale.reshape(1), torch.float32)
    return scale.reshape(1)


def _contrastive_forward(embeddings_a: Tensor, embeddings_b: Tensor, logit_scale: Tensor, mask: Optional[Tensor], smoothing: float,
                         red_code: int, for_backward: bool = True):
    """gather + logits + cross entropy.  Returns (out3, logits_a, logits_b, saved-for-backward tuple)."""
    a = _as_f32(embeddings_a, keep_row_stride=not for_backward)  # (the backward kernels take contiguous local features)
    b = _as_f32(embeddings_b, keep_row_stride=not for_backward)
    if a.stride(0) != b.stride(0):
        a, b = a.contiguous(), b.contiguous()
    B, E = a.shape
    buf, rank, world = gather_packed_features(a, b)  # [W*B, 2E]; W=1 without a process group
    a_all, b_all = buf[:, :E], buf[:, E:]
    scale32 = _scale32(logit_scale)
    row_mask = mask.contiguous().view(torch.uint8) if mask is not None else None
    out3, logits_a, logits_b = ops.contrastive_fwd(a, b, a_all, b_all, 2 * E, scale32, label_offset=B * rank, row_mask=row_mask,
                                                   label_smoothing=smoothing, reduction=red_code)
    return out3, logits_a, logits_b, (a, b, buf, scale32, row_mask, rank, world)


def _contrastive_fwd_impl(a: Tensor, b: Tensor, scale32: Tensor, row_mask: Optional[Tensor], smoothing: float, red_code: int):
    """The training forward as ONE dispatcher op (a, b contiguous fp32 [B, E]): packed all-gather + logits + cross entropy.
    -> [out3, logits_a, logits_b, buf (the gathered [W*B, 2E] features, kept for the backward)]"""
    B, E = a.shape
    buf, rank, _world = gather_packed_features(a, b)
    out3, logits_a, logits_b = ops.contrastive_fwd(a, b, buf[:, :E], buf[:, E:], 2 * E, scale32, label_offset=B * rank, row_mask=row_mask,
                                                   label_smoothing=smoothing, reduction=red_code)
    return [out3, logits_a, logits_b, buf]


def _world() -> int:
    from ...utils.distributed import _dist_ready

    return torch.distributed.get_world_size() if _dist_ready() else 1


def _contrastive_fwd_fake(a, b, scale32, row_mask, smoothing, red_code):
    B, E = a.shape
    W = _world()
    return [a.new_empty((3,)), a.new_empty((B, W * B)), a.new_empty((B, W * B)), a.new_empty((W * B, 2 * E))]


def _contrastive_bwd_impl(g3: Tensor, a: Tensor, b: Tensor, buf: Tensor, scale32: Tensor, logits_a: Tensor, logits_b: Tensor,
                          row_mask: Optional[Tensor], rank: int, world: int, mode: int, smoothing: float, red_code: int):
    """-> [grad_a, grad_b, grad_logit_scale].  mode 0: no process group (both matmul operands are the live tensors); 1: GLOBAL with
    world > 1 (reduce-scatter of every rank's gathered-feature gradients); 2: own block of the gathered gradients only (LOCAL, or
    GLOBAL with world == 1); 3: NONE (no gradient through the gathered operands)."""
    B, E = a.shape
    a_all, b_all = buf[:, :E], buf[:, E:]
    add, all_rows, add_all = None, None, False
    if mode == 0:
        all_rows, add_all = (0, B), True
    elif mode == 1:
        _, _, g_all, _ = ops.contrastive_bwd(a, b, a_all, b_all, 2 * E, scale32, logits_a, logits_b, B * rank, row_mask, smoothing,
                                             red_code, g3, None, (0, world * B))
        if torch.distributed.get_backend() == "nccl":  # RCCL
            add = torch.empty((B, 2 * E), dtype=torch.float32, device=a.device)
            torch.distributed.reduce_scatter_tensor(add, g_all)
        else:  # gloo has no reduce-scatter (tests: two processes on one device): all-reduce, keep the own block
            torch.distributed.all_reduce(g_all)
            add = g_all[B * rank:B * (rank + 1)].contiguous()
    elif mode == 2:
        all_rows, add_all = (B * rank, B), True  # own block only (world == 1: that is everything), no communication
    ga, gb, _, gs = ops.contrastive_bwd(a, b, a_all, b_all, 2 * E, scale32, logits_a, logits_b, B * rank, row_mask, smoothing,
                                        red_code, g3, add, all_rows, add_all)
    return [ga, gb, gs.reshape(1)]


contrastive_fwd_op = _define("contrastive_fwd", "(Tensor a, Tensor b, Tensor scale32, Tensor? row_mask, float smoothing, int red_code) -> Tensor[]",
                             _contrastive_fwd_impl, _contrastive_fwd_fake)
contrastive_bwd_op = _define("contrastive_bwd", "(Tensor g3, Tensor a, Tensor b, Tensor buf, Tensor scale32, Tensor logits_a, Tensor logits_b, "
                             "Tensor? row_mask, int rank, int world, int mode, float smoothing, int red_code) -> Tensor[]", _contrastive_bwd_impl,
                             lambda g3, a, b, buf, scale32, la, lb, rm, rank, world, mode, sm, rc: [torch.empty_like(a), torch.empty_like(b), a.new_empty((1,))])


class _ContrastiveFn(torch.autograd.Function):
    """Autograd node of the contrastive loss on the MI355X kernels (first slice).

    forward = one packed all-gather + mmamd_contrastive_fwd; backward = mmamd_contrastive_bwd and, for BackpropType.GLOBAL with
    world > 1, ONE reduce-scatter of the packed [W*B, 2E] gathered-feature gradients (the reference: two autograd all-gathers whose
    backward are two reduce-scatters, utils/distributed.py:47-48).  LOCAL keeps only this rank's block of the gathered gradients
    (no communication), NONE drops them — inside an initialised process group (world size 1 included); without one the reference
    never gathers, both operands of both matmuls are the live tensors, and every backprop_type differentiates through them.  The logits outputs are not differentiable through this node."""

    @staticmethod
    def forward(ctx, embeddings_a, embeddings_b, logit_scale, mask, backprop_type, smoothing, red_code):
        from ...utils.distributed import _dist_ready

        a, b = _as_f32(embeddings_a), _as_f32(embeddings_b)
        scale32 = _scale32(logit_scale)
        row_mask = mask.contiguous().view(torch.uint8) if mask is not None else None
        out3, logits_a, logits_b, buf = contrastive_fwd_op(a, b, scale32, row_mask, smoothing, red_code)
        dist_on = _dist_ready()
        rank = torch.distributed.get_rank() if dist_on else 0
        world = torch.distributed.get_world_size() if dist_on else 1
        if row_mask is None:
            ctx.save_for_backward(a, b, buf, scale32, logits_a, logits_b)
        else:
            ctx.save_for_backward(a, b, buf, scale32, logits_a, logits_b, row_mask)
        if not dist_on:
            # no process group: the reference's _gather_embeddings_and_labels returns embeddings_a / embeddings_b THEMSELVES and never
            # looks at backprop_type (…:31-33), so both matmul operands stay differentiable for GLOBAL, LOCAL and NONE alike
            mode = 0
        elif backprop_type == BackpropType.GLOBAL and world > 1:
            mode = 1  # gradients of every rank's gathered features, then ONE reduce-scatter; this rank's share is added to grad_a / grad_b
        elif backprop_type in (BackpropType.GLOBAL, BackpropType.LOCAL):
            mode = 2
        else:
            mode = 3
        ctx.meta = (rank, world, mode, smoothing, red_code, row_mask is not None, embeddings_a.dtype, embeddings_b.dtype,
                    logit_scale.dtype, tuple(logit_scale.shape))
        ctx.mark_non_differentiable(logits_a, logits_b)
        return out3, logits_a, logits_b

    @staticmethod
    def backward(ctx, g_out3, _g_la, _g_lb):
        rank, world, mode, smoothing, red_code, has_mask, dt_a, dt_b, dt_s, s_shape = ctx.meta
        if has_mask:
            a, b, buf, scale32, logits_a, logits_b, rm = ctx.saved_tensors
        else:
            a, b, buf, scale32, logits_a, logits_b = ctx.saved_tensors
            rm = None
        g3 = g_out3.detach()
        g3 = (g3 if g3.is_contiguous() else g3.contiguous())
        if g3.dtype != torch.float32:
            g3 = g3.float() if torch.compiler.is_compiling() else ops.convert(g3, torch.float32)
        ga, gb, gs = contrastive_bwd_op(g3, a, b, buf, scale32, logits_a, logits_b, rm, rank, world, mode, smoothing, red_code)

        def cast(t, dt):
            if dt == torch.float32:
                return t
            return t.to(dt) if torch.compiler.is_compiling() else ops.convert(t, dt)

        return cast(ga, dt_a), cast(gb, dt_b), cast(gs, dt_s).reshape(s_shape), None, None, None, None


DEFAULT_LOGIT_SCALE = math.log(1 / 0.07)


class ContrastiveLossWithTemperature(nn.Module):
    """Contrastive loss with a temperature parameter, as used in CLIP and FLAVA.

    Args:
        logit_scale (Union[float, nn.Parameter]): log of the learnable temperature (default ln(1/0.07)); an
            nn.Parameter is adopted as is.
        logit_scale_min (Optional[float]): log of the minimum temperature (default ln 1); None = no lower clamp.
        logit_scale_max (Optional[float]): log of the maximum temperature (default ln 100); None = no upper clamp.

    Inputs: embeddings_a, embeddings_b ([B,E]), backprop_type, cross_entropy_kwargs, mask — as the functional.
    """

    def __init__(self, logit_scale: Union[float, nn.Parameter] = DEFAULT_LOGIT_SCALE,
                 logit_scale_min: Optional[float] = math.log(1), logit_scale_max: Optional[float] = math.log(100)):
        super().__init__()
        torch._C._log_api_usage_once(f"torchmultimodal.{self.__class__.__name__}")
        # same truthiness test as the reference (…:172-175): a 0.0 minimum counts as "not set" there too
        if not logit_scale_min and not logit_scale_max:
            raise ValueError("Only one of `logit_scale_min` and `logit_scale_max` can be None.")
        self.logit_scale_min = logit_scale_min
        self.logit_scale_max = logit_scale_max
        if isinstance(logit_scale, nn.Parameter):
            self.logit_scale = logit_scale
        else:
            self.logit_scale = nn.Parameter(logit_scale * torch.ones([]))

    def forward(self, embeddings_a: Tensor, embeddings_b: Tensor, backprop_type: BackpropType = BackpropType.GLOBAL,
                cross_entropy_kwargs: Optional[Dict[str, Any]] = None, mask: Optional[Tensor] = None) -> Tensor:
        if self.logit_scale.dtype != torch.float32:
            raise ops.MmamdError("logit_scale must be kept in float32")
        # the one sanctioned parameter mutation of the path: in-place clamp before every forward
        if torch.compiler.is_compiling():  # the dispatcher op over the same kernel (csrc/torch_ops.cpp; declared as mutating its argument)
            with torch.no_grad():
                torch.ops.mmamd.clamp_scalar_(self.logit_scale.view(1), self.logit_scale_min, self.logit_scale_max)
        else:
            ops.clamp_scalar_(self.logit_scale.data.view(1), self.logit_scale_min, self.logit_scale_max)
        return contrastive_loss_with_temperature(
            embeddings_a=embeddings_a, embeddings_b=embeddings_b, logit_scale=self.logit_scale,
            backprop_type=backprop_type, cross_entropy_kwargs=cross_entropy_kwargs, mask=mask).loss
