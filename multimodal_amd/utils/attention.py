"""Mask-shape helpers with the contract of torchmultimodal/utils/attention.py:13-64 (host-side tensor plumbing at construction / call time; the kernels
take key masks and [q, k] masks as uint8, see modules/layers/attention.py)."""
from typing import Optional

import torch
from torch import Tensor

# how many broadcast axes go in front of the key axis for a mask of the given rank: [b, k] -> [b, 1, 1, k]; [b, q, k] -> [b, 1, q, k]; [b, h, q, k] as is
_INSERT_AXES = {2: (1, 1), 3: (1,), 4: ()}


def get_extended_attention_mask(attention_mask: Tensor) -> Tensor:
    """A [batch, key] padding mask or a [batch, query, key] mask as a 4-D mask that broadcasts over heads (and queries); 4-D masks pass through.
    Same dtype as the input; any other rank raises ValueError with the reference's message."""
    rank = attention_mask.dim()
    if rank not in _INSERT_AXES:
        raise ValueError("Wrong shape for attention_mask (shape {})".format(attention_mask.shape))
    out = attention_mask
    for _ in _INSERT_AXES[rank]:
        out = out.unsqueeze(1)
    return out.to(dtype=attention_mask.dtype)


def get_causal_attention_mask(tgt_seq_len: int, src_seq_len: Optional[int] = None) -> Tensor:
    """Lower-triangular ones, [target length, source length] (source length defaults to the target's): position t may attend to sources <= t."""
    cols = tgt_seq_len if src_seq_len is None else src_seq_len
    return torch.ones(tgt_seq_len, cols).tril_()
