"""Host helpers mirrored from torchmultimodal/utils/common.py that the path's API needs
(ModelOutput :122-139, load_module_from_url :99-107 without the iopath dependency)."""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import fields
from typing import Any

import torch
from torch import nn


class ModelOutput(OrderedDict):
    def keys(self) -> Any:
        for field in fields(self):  # type: ignore
            yield field.name

    def __getitem__(self, key: Any) -> Any:
        return getattr(self, key)

    def __iter__(self) -> Any:
        yield from self.keys()

    def values(self) -> Any:
        for field in fields(self):  # type: ignore
            yield getattr(self, field.name)

    def items(self) -> Any:
        for field in fields(self):  # type: ignore
            yield field.name, getattr(self, field.name)


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
