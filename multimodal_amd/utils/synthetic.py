"""Seeded synthetic batches for the path: used by tests/, bench.py, smoke() and the golden
generator so that every leg (reference, oracle, HIP) sees bit-identical inputs."""
from __future__ import annotations

from typing import Tuple

import torch

SOT, EOT = 49406, 49407


def clip_batch(batch: int, image_size: int = 224, context_length: int = 77, rank: int = 0,
               vocab_size: int = 49408) -> Tuple[torch.Tensor, torch.Tensor]:
    """images fp32 [B,3,H,W] ~ N(0,1); token ids int64 [B,ctx]: SOT first, one EOT planted at a random
    position >= 1 (argmax is then unambiguous), other ids uniform in [1, EOT-1)."""
    g = torch.Generator().manual_seed(1234 + rank)
    images = torch.randn(batch, 3, image_size, image_size, generator=g)
    hi = min(EOT - 1, vocab_size - 2)
    ids = torch.randint(1, hi, (batch, context_length), generator=g)
    ids[:, 0] = min(SOT, vocab_size - 2)
    pos = torch.randint(1, context_length, (batch,), generator=g)
    ids[torch.arange(batch), pos] = min(EOT, vocab_size - 1)
    return images, ids
