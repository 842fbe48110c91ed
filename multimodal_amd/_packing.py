"""Cache of kernel-ready copies of module parameters.

The kernels want GEMM weights in bf16 and every small vector (biases, LayerNorm affine, embeddings tables'
positional rows, projections of the pooled row) in fp32, whatever dtype the nn.Parameters are kept in.
Conversions run on the HIP convert kernel (ops.convert) and are cached per parameter; a cached copy is
invalidated when the parameter's storage pointer or in-place version counter changes (load_state_dict,
optimizer.step, .to(...)).  Parameters that already have the wanted dtype are used in place (no copy).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import ops


class PackedCache:
    def __init__(self) -> None:
        self._store: Dict[Tuple[int, torch.dtype], Tuple[int, int, torch.device, torch.Tensor]] = {}

    def get(self, p: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        t = p.detach()
        if not t.is_cuda:
            raise ops.MmamdError(
                f"parameter lives on {t.device}: move the module to a HIP device (.to('cuda')); there is no CPU path")
        if t.dtype == dtype and t.is_contiguous():
            return t
        key = (id(p), dtype)
        hit = self._store.get(key)
        if hit is not None and hit[0] == t.data_ptr() and hit[1] == p._version and hit[2] == t.device:
            return hit[3]
        src = t if t.is_contiguous() else t.contiguous()
        conv = ops.convert(src, dtype)
        self._store[key] = (t.data_ptr(), p._version, t.device, conv)
        return conv

    def clear(self) -> None:
        self._store.clear()
