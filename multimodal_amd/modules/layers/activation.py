"""Host-side mirror of torchmultimodal/modules/layers/activation.py:12-25 (SiLU = QuickGELU).

On the hot path this activation never runs as its own op: it is the MMAMD_ACT_QUICKGELU epilogue of the
MLP up-projection GEMM (csrc/gemm.hip).  The module exists so `activation=SiLU()` configuration code and
isinstance checks keep working.
"""
from torch import nn, Tensor


class SiLU(nn.Module):
    r"""Sigmoid Linear Unit  SiLU(x) = x * sigmoid(1.702 * x)  (QuickGELU of the CLIP paper)."""

    coefficient = 1.702

    def forward(self, x: Tensor) -> Tensor:
        raise NotImplementedError(
            "SiLU is fused into the GEMM epilogue on the MI355X path (MMAMD_ACT_QUICKGELU); "
            "a standalone elementwise launch is not part of the hot path")
