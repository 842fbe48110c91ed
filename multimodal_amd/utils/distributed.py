"""Feature all-gather for the global contrastive loss — host-side mirror of
torchmultimodal/utils/distributed.py:16-90 (BackpropType, gather_tensor, concat_gather_all_gpu, get_rank)
plus the packed single-collective gather the MI355X loss path uses.

The collective backend is torch.distributed: on ROCm the "nccl" backend IS RCCL, which runs the 8-GPU
all-gather over the xGMI links of one node; on CPU (tests) it is gloo.  The path's message is tiny
(<= 768 KiB per rank) and latency-bound, so the product path issues ONE all_gather_into_tensor of a packed
[B, 2E] (image | text) block straight into the contiguous [W*B, 2E] buffer the loss kernel reads with a
row stride — instead of the reference's two list-output all_gathers + two torch.cat copies
(contrastive_loss_with_temperature.py:35-47).
"""
from __future__ import annotations

from enum import Enum
from typing import List, Tuple

import torch
from torch import Tensor


class BackpropType(Enum):
    """How to backpropagate gradients during all-gather op. GLOBAL will backpropagate
    to all workers, LOCAL to only the current worker, and NONE will not backpropagate
    at all.  (reference: utils/distributed.py:16-25)"""

    GLOBAL = 0
    LOCAL = 1
    NONE = 2


def _dist_ready() -> bool:
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def get_rank() -> int:
    """reference: utils/distributed.py:86-90"""
    if _dist_ready():
        return torch.distributed.get_rank()
    return 0


def gather_tensor(tensor: Tensor, backprop_type: BackpropType = BackpropType.GLOBAL) -> List[Tensor]:
    """Gathers a tensor across all GPUs; returns a list of world_size tensors
    (reference: utils/distributed.py:28-58 — same three backprop behaviours)."""
    world_size = torch.distributed.get_world_size()
    if backprop_type == BackpropType.GLOBAL:
        # autograd-aware all_gather (backward = reduce_scatter), as the reference uses for GLOBAL
        from torch.distributed.nn.functional import all_gather as all_gather_with_backprop

        return list(all_gather_with_backprop(tensor))
    tensor_all_gpus = [torch.zeros_like(tensor) for _ in range(world_size)]
    torch.distributed.all_gather(tensor_all_gpus, tensor)
    if backprop_type == BackpropType.LOCAL:
        tensor_all_gpus[get_rank()] = tensor
    return tensor_all_gpus


def concat_gather_all_gpu(tensor: Tensor, backprop_type: BackpropType = BackpropType.GLOBAL, dim: int = 0) -> Tensor:
    """reference: utils/distributed.py:61-83"""
    if not _dist_ready():
        return tensor
    return torch.cat(gather_tensor(tensor, backprop_type), dim=dim)


def gather_packed_features(a: Tensor, b: Tensor) -> Tuple[Tensor, int, int]:
    """One collective for both modalities (forward path).

    a, b: [B, E] same dtype/device.  Returns (buf [W*B, 2E], rank, world_size) where
    buf[r*B:(r+1)*B, :E] is rank r's `a` and buf[r*B:(r+1)*B, E:] its `b`.  Without an initialised
    process group this is just the packed local block (W = 1).
    """
    if a.shape != b.shape or a.dim() != 2:
        raise ValueError(f"expected two [B,E] tensors of equal shape, got {tuple(a.shape)} and {tuple(b.shape)}")
    B, E = a.shape
    if (a.dtype == b.dtype and a.device == b.device and a.stride() == (2 * E, 1) and b.stride() == (2 * E, 1)
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and b.storage_offset() == a.storage_offset() + E
            and a.untyped_storage().nbytes() >= (a.storage_offset() + B * 2 * E) * a.element_size()):
        # a and b already ARE the two halves of one packed [B, 2E] block (CLIP.forward normalises straight into it): no packing copies
        packed = a.as_strided((B, 2 * E), (2 * E, 1), a.storage_offset())
    else:
        packed = torch.empty((B, 2 * E), dtype=a.dtype, device=a.device)
        packed[:, :E].copy_(a)  # device memcpy: packing the message, not compute
        packed[:, E:].copy_(b)
    if not _dist_ready():
        return packed, 0, 1
    world = torch.distributed.get_world_size()
    buf = torch.empty((world * B, 2 * E), dtype=a.dtype, device=a.device)
    torch.distributed.all_gather_into_tensor(buf, packed)
    return buf, torch.distributed.get_rank(), world
