"""Host helpers mirrored from torchmultimodal/utils/common.py that the path's API needs
(ModelOutput :122-139, load_module_from_url :99-107 without the iopath dependency)."""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import fields
from typing import Any

import torch
from torch import nn


class ModelOutput(OrderedDict):
    """Base of the dataclass outputs that must also read like a mapping (contract of the reference's utils/common.py:122-139): iteration and `keys()` give
    the dataclass field names in declaration order, `values()` / `items()` / `out[name]` read the ATTRIBUTES -- the dict storage itself stays empty."""

    def _field_names(self):
        return [f.name for f in fields(self)]  # type: ignore[arg-type]

    def keys(self) -> Any:
        return iter(self._field_names())

    def __iter__(self) -> Any:
        return self.keys()

    def __getitem__(self, key: Any) -> Any:
        return getattr(self, key)

    def values(self) -> Any:
        return (getattr(self, name) for name in self._field_names())

    def items(self) -> Any:
        return ((name, getattr(self, name)) for name in self._field_names())


def load_module_from_url(model: nn.Module, url: str, strict: bool = True, progress: bool = True) -> None:
    """Load a published checkpoint (same state_dict keys as the reference).

    `url` may be a local path or an http(s) URL (fetched with torch.hub's downloader; needs network)."""
    if url.startswith(("http://", "https://")):
        state_dict = torch.hub.load_state_dict_from_url(url, map_location="cpu", progress=progress)
    else:
        state_dict = torch.load(url, map_location="cpu")
    model.load_state_dict(state_dict, strict=strict)
    from .._packing import invalidate_packed

    invalidate_packed(model)  # kernel-ready copies of the old weights (bf16 / fp32 packs) are rebuilt on the next forward
