"""Host-side mirror of torchmultimodal/modules/layers/mlp.py:13-66.

Same constructor, same `model` nn.Sequential (so state_dict keys are `model.0.weight`, `model.2.weight`, ... and a seeded
construction consumes the RNG identically).  forward() does not run the Sequential: every Linear is one bf16 MFMA GEMM
(csrc/gemm.hip) with the bias and the FOLLOWING activation fused into its epilogue.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Union

import torch
from torch import nn

from ... import ops
from ..._packing import PackedCache
from .activation import SiLU


def fused_activation_code(mod: nn.Module) -> Optional[int]:
    """GEMM epilogue code of an activation module, None if the kernels do not have it."""
    if isinstance(mod, nn.GELU) and getattr(mod, "approximate", "none") == "none":
        return ops.ACT_GELU_ERF
    if isinstance(mod, SiLU):
        return ops.ACT_QUICKGELU
    return None


class MLP(nn.Module):
    def __init__(
        self,
        in_dim: int,
        out_dim: int,
        hidden_dims: Optional[Union[int, List[int]]] = None,
        dropout: float = 0.5,
        activation: Callable[..., nn.Module] = nn.ReLU,
        normalization: Optional[Callable[..., nn.Module]] = None,
    ) -> None:
        super().__init__()
        layers = nn.ModuleList()
        if hidden_dims is None:
            hidden_dims = []
        if isinstance(hidden_dims, int):
            hidden_dims = [hidden_dims]
        for hidden_dim in hidden_dims:
            layers.append(nn.Linear(in_dim, hidden_dim))
            if normalization:
                layers.append(normalization(hidden_dim))
            layers.append(activation())
            if dropout > 0:
                layers.append(nn.Dropout(dropout))
            in_dim = hidden_dim
        layers.append(nn.Linear(in_dim, out_dim))
        self.model = nn.Sequential(*layers)
        self._packed = PackedCache()

    def plan(self):
        """[(linear, activation code)] — raises for module sequences the GEMM epilogues cannot express."""
        mods = list(self.model)
        steps, i = [], 0
        while i < len(mods):
            lin = mods[i]
            if not isinstance(lin, nn.Linear):
                raise ops.MmamdError(f"MLP on the MI355X path: unsupported layer {type(lin).__name__} (normalization inside "
                                     "the MLP is not on the contrastive path)")
            act = ops.ACT_NONE
            i += 1
            if i < len(mods) and not isinstance(mods[i], (nn.Linear, nn.Dropout)):
                code = fused_activation_code(mods[i])
                if code is None:
                    raise ops.MmamdError(f"MLP on the MI355X path: activation {type(mods[i]).__name__} has no fused GEMM "
                                         "epilogue (nn.GELU and the CLIP SiLU/QuickGELU do)")
                act = code
                i += 1
            if i < len(mods) and isinstance(mods[i], nn.Dropout):
                if self.training and mods[i].p > 0:
                    raise ops.MmamdError("MLP on the MI355X path: dropout > 0 in training mode is not implemented")
                i += 1
            steps.append((lin, act))
        return steps

    def run(self, h: torch.Tensor, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """h: bf16 [M, in_dim].  Returns fp32 [M, out_dim] (+ residual, which may alias out)."""
        steps = self.plan()
        pk = self._packed.get
        for n, (lin, act) in enumerate(steps):
            last = n == len(steps) - 1
            b = pk(lin.bias, torch.float32) if lin.bias is not None else None
            if last:
                h = ops.gemm_bf16(h, pk(lin.weight, torch.bfloat16), b, act=act, residual=residual, out_dtype=torch.float32, out=out)
            else:
                h = ops.gemm_bf16(h, pk(lin.weight, torch.bfloat16), b, act=act)
        return h

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        xc = x if x.is_contiguous() else x.contiguous()
        h = ops.convert(xc.view(-1, xc.shape[-1]), torch.bfloat16)
        y = self.run(h)
        return y.view(*x.shape[:-1], y.shape[-1])
