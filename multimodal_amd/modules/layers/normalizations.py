"""Host-side mirror of torchmultimodal/modules/layers/normalizations.py:13-25 (Fp32LayerNorm)."""
from typing import Any

from torch import nn, Tensor

import torch

from ... import _torch_ops, ops
from ..._packing import PackedCache

_torch_ops.try_load()


class Fp32LayerNorm(nn.LayerNorm):
    """LayerNorm whose statistics are computed in fp32 whatever the input dtype; result cast back to the
    input dtype.  forward() launches the wave-per-row HIP kernel (csrc/rowops.hip: layernorm_kernel)."""

    def __init__(self, *args: Any, **kwargs: Any) -> None:
        super().__init__(*args, **kwargs)
        self._packed = PackedCache()

    def forward(self, x: Tensor) -> Tensor:
        if torch.jit.is_scripting():  # dispatcher op (csrc/torch_ops.cpp): the same kernel, visible to TorchScript
            return torch.ops.mmamd.layernorm(x.contiguous(), self.weight, self.bias, self.eps, 0 if x.dtype == torch.float32 else 1)
        else:
            return self._forward_host(x)

    def _ln_ops(self, x: Tensor, out_dtype: int) -> Tensor:
        """LayerNorm of fp32 / bf16 rows through the dispatcher op with an explicit output dtype code (0 = fp32, 1 = bf16: the operand of
        the next GEMM) — for the scripted forwards of the enclosing layers (`eps` is a TorchScript constant of THIS module)."""
        return torch.ops.mmamd.layernorm(x, self.weight, self.bias, self.eps, out_dtype)

    @torch.jit.unused
    def _forward_host(self, x: Tensor) -> Tensor:
        if self.weight is None or self.bias is None or len(self.normalized_shape) != 1:
            raise ops.MmamdError("Fp32LayerNorm on the MI355X path needs a 1-D affine LayerNorm")
        if torch.compiler.is_compiling():
            return torch.ops.mmamd.layernorm(x.contiguous(), self.weight, self.bias, self.eps, 0 if x.dtype == torch.float32 else 1)
        if torch.is_grad_enabled() and (x.requires_grad or (self.training and self.weight.requires_grad)):
            from ..._autograd import LayerNormFn  # differentiable path: forward and backward HIP kernels

            return LayerNormFn.apply(x, self.weight, self.bias, self.eps)
        g = self._packed.get(self.weight, torch.float32)
        b = self._packed.get(self.bias, torch.float32)
        xc = x if x.is_contiguous() else x.contiguous()
        return ops.layernorm(xc, g, b, self.eps, out_dtype=x.dtype)
