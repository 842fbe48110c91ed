"""How the layers of a model are mapped to kernel launches: ONE settings object, resolved once (import time, from the environment) and
changed only through `set_schedule` — the forwards read it, they do not look at environment variables themselves.

Every choice is between paths that compute the same function; the defaults are the measured-fastest ones on MI355X.

    two_tower   "auto" | "grouped" | "streams"    CLIP pair: layer-locked grouped launches on one stream vs one stream per tower
                                                  (auto: grouped only where it measured faster, _transformer.two_stacks_groupable)
    residual    "epilogue" | "delta_ln"           x += proj(...) inside the GEMM epilogue (fp32 read-modify-write per tile), or the GEMM
                                                  stores a bf16 delta and ONE streaming kernel does x += delta; hn = LN(x) for both towers
    side_stream  True | False                     tower-agnostic path: second tower on a side stream (False = one stream)
    flava_batched_passes  True | False            FLAVA inference: the unmasked and the masked pass of a tower as ONE pass over a 2B batch
    train_side_stream  True | False               CLIP training step: the text tower's forward (and, through autograd, backward) on a side stream
                                                  (same kernels, bit-identical step; -1.3 ... -4 ms of 54 depending on the box)

Environment (read once): MMAMD_TWO_TOWER, MMAMD_RESIDUAL (epilogue | delta_ln), MMAMD_SINGLE_STREAM=1.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, replace

_TWO_TOWER = ("auto", "grouped", "streams")
_RESIDUAL = ("epilogue", "delta_ln")


@dataclass(frozen=True)
class Schedule:
    two_tower: str = "auto"
    residual: str = "epilogue"
    side_stream: bool = True
    flava_batched_passes: bool = True
    train_side_stream: bool = True

    def __post_init__(self):
        if self.two_tower not in _TWO_TOWER:
            raise ValueError(f"two_tower must be one of {_TWO_TOWER}, got {self.two_tower!r}")
        if self.residual not in _RESIDUAL:
            raise ValueError(f"residual must be one of {_RESIDUAL}, got {self.residual!r}")


def _from_env() -> Schedule:
    return Schedule(two_tower=os.environ.get("MMAMD_TWO_TOWER", "auto"), residual=os.environ.get("MMAMD_RESIDUAL", "epilogue"),
                    side_stream=os.environ.get("MMAMD_SINGLE_STREAM") != "1")


_current = _from_env()


def get_schedule() -> Schedule:
    return _current


def set_schedule(**changes) -> Schedule:
    """Replace fields of the process-wide schedule (tools / tests: A-B runs in one process).  Returns the previous schedule."""
    global _current
    prev = _current
    _current = replace(_current, **changes)
    return prev
