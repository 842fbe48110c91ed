"""Host-side mirror of torchmultimodal/transforms/text_transforms.py:14-217 (Truncate, AddToken, PadTransform, ToTensor and the
truncate / add_token / to_tensor functions).  Pure host logic on Python lists; the tensors they build are CPU int64, as in the
reference -- the batch goes to the device as one [B, L] copy by the caller (CLIPTextTransform(device=...))."""
from __future__ import annotations

from typing import Any, List, Optional, Union

import numpy as np
import torch
from torch import nn, Tensor


def _is_batch(x: Any) -> bool:
    """A list whose elements are lists is a batch of sequences; anything else that is a list is one sequence."""
    if not isinstance(x, list):
        raise TypeError("Input type not supported")
    nested = [isinstance(e, list) for e in x]
    if any(nested) and not all(nested):
        raise TypeError("Input type not supported")
    return bool(nested) and nested[0]


def _check_elems(x: List[Any], kinds) -> None:
    for e in x:
        if isinstance(e, bool) or not isinstance(e, kinds):
            raise TypeError("Input type not supported")


def truncate(input: Any, max_seq_len: int) -> Any:
    """text_transforms.py:132-158."""
    if _is_batch(input):
        for ids in input:
            _check_elems(ids, (int, str))
        return [ids[:max_seq_len] for ids in input]
    _check_elems(input, (int, str))
    return input[:max_seq_len]


def add_token(input: Any, token_id: Any, begin: bool = True) -> Any:
    """text_transforms.py:161-217: the token and the sequence elements must be of one kind (int ids or str tokens)."""
    kind = int if isinstance(token_id, int) and not isinstance(token_id, bool) else str if isinstance(token_id, str) else None
    if kind is None:
        raise TypeError("Input type not supported")
    if _is_batch(input):
        for ids in input:
            _check_elems(ids, kind)
        return [[token_id] + ids for ids in input] if begin else [ids + [token_id] for ids in input]
    _check_elems(input, kind)
    return [token_id] + input if begin else input + [token_id]


def to_tensor(input: Any, padding_value: Optional[int] = None, dtype: torch.dtype = torch.long) -> Tensor:
    """text_transforms.py:101-129: one sequence -> 1-D int64; a batch -> [B, longest] (ragged batches need a padding_value)."""
    if not _is_batch(input):
        _check_elems(input, int)
        return torch.tensor(input, dtype=torch.long)
    for ids in input:
        _check_elems(ids, int)
    if padding_value is None:
        return torch.tensor(input, dtype=dtype)
    longest = max((len(ids) for ids in input), default=0)
    buf = np.full((len(input), longest), int(padding_value), dtype=np.int64)
    for i, ids in enumerate(input):
        buf[i, : len(ids)] = ids
    out = torch.from_numpy(buf)
    return out if dtype == torch.long else out.to(dtype)


class Truncate(nn.Module):
    def __init__(self, max_seq_len: int) -> None:
        super().__init__()
        self.max_seq_len = max_seq_len

    def forward(self, x: Any) -> Any:
        return truncate(x, self.max_seq_len)


class AddToken(nn.Module):
    def __init__(self, token: Union[int, str], begin: bool = True) -> None:
        super().__init__()
        self.token = token
        self.begin = begin

    def forward(self, input: Any) -> Any:
        return add_token(input, self.token, self.begin)


class PadTransform(nn.Module):
    """Right-pads the last dimension to max_length (text_transforms.py:57-81); longer inputs pass through."""

    def __init__(self, max_length: int, pad_value: int) -> None:
        super().__init__()
        self.max_length = max_length
        self.pad_value = float(pad_value)

    def forward(self, x: Tensor) -> Tensor:
        n = x.size(-1)
        if n < self.max_length:
            x = torch.nn.functional.pad(x, (0, self.max_length - n), value=self.pad_value)
        return x


class ToTensor(nn.Module):
    def __init__(self, padding_value: Optional[int] = None, dtype: torch.dtype = torch.long) -> None:
        super().__init__()
        self.padding_value = padding_value
        self.dtype = dtype

    def forward(self, input: Any) -> Tensor:
        return to_tensor(input, padding_value=self.padding_value, dtype=self.dtype)
