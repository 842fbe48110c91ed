"""Host-side mirror of torchmultimodal/utils/attention.py:13-64 (mask shape helpers; construction-time tensor plumbing)."""
from typing import Optional

import torch
from torch import Tensor


def get_extended_attention_mask(attention_mask: Tensor) -> Tensor:
    """Makes attention masks broadcastable along head and sequence dimensions ([b,s] -> [b,1,1,s]; [b,q,k] -> [b,1,q,k])."""
    if attention_mask.dim() == 4:
        extended_attention_mask = attention_mask
    elif attention_mask.dim() == 3:
        extended_attention_mask = attention_mask[:, None, :, :]
    elif attention_mask.dim() == 2:
        extended_attention_mask = attention_mask[:, None, None, :]
    else:
        raise ValueError("Wrong shape for attention_mask (shape {})".format(attention_mask.shape))
    return extended_attention_mask.to(dtype=attention_mask.dtype)


def get_causal_attention_mask(tgt_seq_len: int, src_seq_len: Optional[int] = None) -> Tensor:
    """Causal attention mask of dimensions (target_sequence_length, source_sequence_length)."""
    if src_seq_len is None:
        src_seq_len = tgt_seq_len
    return torch.tril(torch.ones(tgt_seq_len, src_seq_len))
