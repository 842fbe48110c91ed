"""Host-side mirror of torchmultimodal/modules/layers/activation.py:12-25 (SiLU = QuickGELU).

On the hot path this activation never runs as its own op: it is the MMAMD_ACT_QUICKGELU epilogue of the
MLP up-projection GEMM (csrc/gemm.hip).  Called on its own (the reference's KAT: tests/modules/layers/test_activation.py:12-16,
silu(1) = 0.8458; an `activation=SiLU()` inside user code) it is one elementwise launch, `mmamd_activation`, differentiable
through its own backward launch.
"""
import torch
from torch import nn, Tensor

from ... import _torch_ops, ops

_torch_ops.try_load()


class _ActivationFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, act: int) -> Tensor:
        xc = x.detach()
        xc = xc if xc.is_contiguous() else xc.contiguous()
        ctx.save_for_backward(xc)
        ctx.act = act
        return ops.activation(xc, act)

    @staticmethod
    def backward(ctx, dy: Tensor):
        (xc,) = ctx.saved_tensors
        d = dy.detach()
        d = d if d.is_contiguous() else d.contiguous()
        if d.dtype != xc.dtype:
            d = ops.convert(d, xc.dtype)
        return ops.activation(xc, ctx.act, dy=d), None


class SiLU(nn.Module):
    r"""Sigmoid Linear Unit  SiLU(x) = x * sigmoid(1.702 * x)  (QuickGELU of the CLIP paper).

    .. math:: \text{SiLU}(x) = x * \sigma(1.702 * x)

    Shape: input (*) -> output (*), fp32 or bf16 on a HIP device.
    """

    coefficient = 1.702

    def forward(self, x: Tensor) -> Tensor:
        if torch.jit.is_scripting():
            return torch.ops.mmamd.activation(x.contiguous(), 1)  # 1 = MMAMD_ACT_QUICKGELU
        else:
            return self._forward_host(x)

    @torch.jit.unused
    def _forward_host(self, x: Tensor) -> Tensor:
        if torch.compiler.is_compiling() and not (torch.is_grad_enabled() and x.requires_grad):
            return torch.ops.mmamd.activation(x.contiguous(), 1)
        return _ActivationFn.apply(x, ops.ACT_QUICKGELU)
