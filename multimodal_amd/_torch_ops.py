"""Loader of the TORCH_LIBRARY(mmamd, ...) shim (csrc/torch_ops.cpp -> lib/libmmamd_torch.so): `torch.ops.mmamd.*`.

The dispatcher ops are what the scriptable / compilable forwards call (`torch.jit.script`, `torch.compile`); the eager forwards keep
the ctypes binding (multimodal_amd/ops.py), which needs no torch C++ ABI.  Both end in the same extern "C" entry points of
include/mmamd.h.  Like the ctypes binding there is no fallback: a missing library raises.
"""
from __future__ import annotations

import torch

from ._lib import BF16, F32, MmamdError  # noqa: F401  (dtype codes of the `int dtype` op arguments)

_LOADED = False


def try_load() -> bool:
    """Register the ops if the shim is there (or can be built); False instead of an exception otherwise.  Called when the modules with
    scriptable forwards are imported: `torch.jit.script` resolves `torch.ops.mmamd.*` at compile time."""
    try:
        load()
        return True
    except Exception:  # noqa: BLE001
        return False


def loaded() -> bool:
    """True once the shim has been registered in this process."""
    return bool(_LOADED)


def load():
    """Register the ops (once) and return the `torch.ops.mmamd` namespace."""
    global _LOADED
    if not _LOADED:
        from . import build

        path = build.TORCH_LIB
        if not path.exists():
            try:
                path = build.build_torch_ops()
            except Exception as e:  # noqa: BLE001
                raise MmamdError(f"{build.TORCH_LIB} is not built and could not be built here ({e}); run `python -m multimodal_amd.build`") from e
        try:
            torch.ops.load_library(str(path))
        except OSError as e:
            raise MmamdError(f"cannot load {path}: {e}") from e
        _LOADED = True
    return torch.ops.mmamd
