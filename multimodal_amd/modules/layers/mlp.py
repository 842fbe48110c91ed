"""Host-side mirror of torchmultimodal/modules/layers/mlp.py:13-66.

Same constructor, same `model` nn.Sequential (so state_dict keys are `model.0.weight`, `model.2.weight`, ... and a seeded
construction consumes the RNG identically).  forward() does not run the Sequential: every Linear is one bf16 MFMA GEMM
(csrc/gemm.hip) with the bias and the FOLLOWING activation fused into its epilogue.
"""

from typing import Callable, List, Optional, Union

import torch
from torch import nn

from ... import _torch_ops, ops
from ..._packing import PackedCache
from .activation import SiLU

_torch_ops.try_load()


ACT_RELU_EXACT = -1  # plan() code of nn.ReLU: no GEMM epilogue has it; MLPs that use it (classifier heads) run the exact-fp32 row path


def fused_activation_code(mod: nn.Module) -> Optional[int]:
    """GEMM epilogue code of an activation module, None if the kernels do not have it."""
    if isinstance(mod, nn.GELU) and getattr(mod, "approximate", "none") == "none":
        return ops.ACT_GELU_ERF
    if isinstance(mod, SiLU):
        return ops.ACT_QUICKGELU
    if isinstance(mod, nn.ReLU):
        return ACT_RELU_EXACT
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
        # what the scripted forward needs to know statically: the GEMM-epilogue code of the activation (-2: none of the kernels', -1: nn.ReLU
        # = exact-fp32 row path, eager only; -3: a normalization inside the MLP) and the number of Linear layers
        code = fused_activation_code(activation()) if len(hidden_dims) > 0 else ops.ACT_NONE
        self._act_code: int = -3 if normalization else (-2 if code is None else int(code))
        self._n_linear: int = len(hidden_dims) + 1

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
                    raise ops.MmamdError(f"MLP on the MI355X path: activation {type(mods[i]).__name__} has no kernel "
                                         "(nn.GELU and the CLIP SiLU/QuickGELU are GEMM epilogues, nn.ReLU runs on the exact-fp32 row path)")
                act = code
                i += 1
            if i < len(mods) and isinstance(mods[i], nn.Dropout):  # applied by the training paths (hidden_dropout_p); identity in eval mode
                i += 1
            steps.append((lin, act))
        return steps

    def hidden_dropout_p(self) -> float:
        """p of the nn.Dropout behind every hidden activation (reference mlp.py:59-60; one value per MLP), 0.0 if there is none."""
        ps = {float(m.p) for m in self.model if isinstance(m, nn.Dropout)}
        if len(ps) > 1:
            raise ops.MmamdError("MLP on the MI355X path: one dropout rate per MLP")
        return ps.pop() if ps else 0.0

    def run(self, h: torch.Tensor, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """h: bf16 [M, in_dim].  Returns fp32 [M, out_dim] (+ residual, which may alias out)."""
        steps = self.plan()
        if any(act == ACT_RELU_EXACT for _, act in steps):
            raise ops.MmamdError("MLP.run: nn.ReLU MLPs take the exact-fp32 row path (call the module, not run())")
        pk = self._packed.get
        for n, (lin, act) in enumerate(steps):
            last = n == len(steps) - 1
            b = pk(lin.bias, torch.float32) if lin.bias is not None else None
            if last:
                h = ops.gemm_bf16(h, pk(lin.weight, torch.bfloat16), b, act=act, residual=residual, out_dtype=torch.float32, out=out)
            else:
                h = ops.gemm_bf16(h, pk(lin.weight, torch.bfloat16), b, act=act)
        return h

    # rows up to which an MLP with nn.ReLU (or no) activations runs in exact fp32 (mmamd_rows_linear_f32): classifier heads see one
    # row per sample
    EXACT_ROWS_MAX = 8192

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if torch.jit.is_scripting():
            return self._forward_ops(x)
        else:
            return self._forward_host(x)

    def _run_ops(self, h: torch.Tensor, residual: Optional[torch.Tensor]) -> torch.Tensor:
        """run() through the dispatcher ops (torch.ops.mmamd.*): h bf16 [M, in_dim] -> fp32 [M, out_dim] (+ residual)."""
        if self._act_code < 0:
            raise RuntimeError("scripted MLP on the MI355X path: nn.GELU / SiLU MLPs without normalization only (nn.ReLU heads use the eager forward)")
        i = 0
        for m in self.model:
            if hasattr(m, "weight"):  # the Linear layers (statically resolved per module of the Sequential)
                i += 1
                if i < self._n_linear:
                    h = torch.ops.mmamd.gemm_bf16(h, m.weight, m.bias, None, self._act_code, 1)
                else:
                    h = torch.ops.mmamd.gemm_bf16(h, m.weight, m.bias, residual, 0, 0)
        return h

    def _forward_ops(self, x: torch.Tensor) -> torch.Tensor:
        d = x.size(-1)
        h = torch.ops.mmamd.convert(x.contiguous().view(-1, d), 1)
        y = self._run_ops(h, None)
        shape = x.size()[:-1] + [y.size(1)]
        return y.view(shape)

    @torch.jit.unused
    def _forward_host(self, x: torch.Tensor) -> torch.Tensor:
        if torch.compiler.is_compiling() and self._act_code >= 0 and not (torch.is_grad_enabled() and x.requires_grad):
            return self._forward_ops(x)
        steps = self.plan()
        xc = x if x.is_contiguous() else x.contiguous()
        rows = xc.view(-1, xc.shape[-1])
        if all(act in (ops.ACT_NONE, ACT_RELU_EXACT) for _, act in steps) and any(act == ACT_RELU_EXACT for _, act in steps):
            if rows.shape[0] > self.EXACT_ROWS_MAX:
                raise ops.MmamdError(f"MLP with nn.ReLU on the MI355X path is the exact-fp32 classifier-head form (<= {self.EXACT_ROWS_MAX} "
                                     f"rows, got {rows.shape[0]}); token-level MLPs use nn.GELU / SiLU (fused GEMM epilogues)")
            if rows.dtype != torch.float32:
                raise ops.MmamdError("MLP (exact-fp32 row path) takes fp32 activations")
            from ..._autograd import SmallLinearF32Fn, dropout_train, wants_grad

            h = rows
            if wants_grad(self) or (torch.is_grad_enabled() and x.requires_grad):
                pdrop = self.hidden_dropout_p() if self.training else 0.0
                for n, (lin, act) in enumerate(steps):
                    h = SmallLinearF32Fn.apply(h, lin.weight, lin.bias, act == ACT_RELU_EXACT)
                    if pdrop > 0 and n + 1 < len(steps):  # Linear -> activation -> Dropout per hidden layer (reference mlp.py:52-61)
                        h = dropout_train(h, pdrop, site=n)
            else:
                pk = self._packed.get
                for lin, act in steps:
                    b = pk(lin.bias, torch.float32) if lin.bias is not None else None
                    h = ops.rows_linear_f32(h, h.shape[1], h.shape[0], pk(lin.weight, torch.float32), b, relu=(act == ACT_RELU_EXACT))
            return h.view(*x.shape[:-1], h.shape[-1])
        if any(act == ACT_RELU_EXACT for _, act in steps):
            raise ops.MmamdError("MLP on the MI355X path: nn.ReLU cannot be mixed with the GEMM-epilogue activations in one MLP")
        from ..._autograd import forbid_detached_forward

        forbid_detached_forward(self, x)
        h = ops.convert(rows, torch.bfloat16)
        y = self.run(h)
        return y.view(*x.shape[:-1], y.shape[-1])
