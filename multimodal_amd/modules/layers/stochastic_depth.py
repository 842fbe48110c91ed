"""Stochastic depth — host-side mirror of torchvision.ops.StochasticDepth(p, mode), which the reference's TransformerEncoderLayer puts on
both residual branches when `drop_path_rate` is given (torchmultimodal/modules/layers/transformer.py:64-67; rates per layer from
torch.linspace(0, drop_path_rate, n_layer), :190-191).  torchvision is not a dependency here.

Training: whole samples ("row") or the whole batch ("batch") of the branch are zeroed with probability p, survivors scaled by 1 / (1 - p); the
decisions come from the Philox generator of csrc/dropout.hip (one per sample), so the backward regenerates them.  Evaluation / p = 0: identity.
Inside a layer stack the encoder's autograd node applies it fused with the residual add (_autograd.stack_drop_spec); this forward serves a
module called on its own.
"""
import torch
from torch import nn, Tensor


class StochasticDepth(nn.Module):
    def __init__(self, p: float, mode: str) -> None:
        super().__init__()
        if p < 0.0 or p > 1.0:
            raise ValueError(f"drop probability has to be between 0 and 1, but got {p}")
        if mode not in ("batch", "row"):
            raise ValueError(f"mode has to be either 'batch' or 'row', but got {mode}")
        self.p = p
        self.mode = mode

    def forward(self, input: Tensor) -> Tensor:
        if not self.training or self.p == 0.0:
            return input
        from ... import ops
        from ..._autograd import dropout_train

        if self.p >= 1.0:
            raise ops.MmamdError("StochasticDepth(p = 1) drops every sample: not meaningful on the MI355X path")
        x = input if input.is_contiguous() else input.contiguous()
        group = x.numel() if self.mode == "batch" else x.numel() // max(x.shape[0], 1)
        return dropout_train(x, self.p, group=group)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(p={self.p}, mode={self.mode})"
