"""Shared test helpers."""
from __future__ import annotations

import numpy as np
import torch


def sd_to_numpy(module: torch.nn.Module):
    return {k: v.detach().cpu().float().numpy() for k, v in module.state_dict().items()}


def fixture_sd(z, prefix="sd."):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


def assert_checksums(module: torch.nn.Module, z) -> None:
    """The seeded re-creation of a full-size model must reproduce the weights the reference produced when the
    fixture was generated (per-tensor sum and abs-sum stored by tests/golden/make_golden.py)."""
    sd = module.state_dict()
    keys = [str(k) for k in z["keys"]]
    assert list(sd.keys()) == keys
    for k, s, a in zip(keys, z["sums"], z["asums"]):
        v = sd[k].double()
        assert abs(float(v.sum()) - float(s)) <= 1e-6 * max(1.0, abs(float(a))), k
        assert abs(float(v.abs().sum()) - float(a)) <= 1e-9 * max(1.0, abs(float(a))), k
