"""How the layers of a model are mapped to kernel launches: ONE settings object, resolved once (import time, from the environment) and
changed only through `set_schedule` — the forwards read it, they do not look at environment variables themselves.

Every choice is between paths that compute the same function; the defaults are the measured-fastest ones on MI355X.

    two_tower   "auto" | "grouped" | "streams"    CLIP pair: layer-locked grouped launches on one stream vs one stream per tower
                                                  (auto: grouped only where it measured faster, _transformer.two_stacks_groupable)
    side_stream  True | False                     tower-agnostic path: second tower on a side stream (False = one stream)
    flava_batched_passes  True | False            FLAVA inference: the unmasked and the masked pass of a tower as ONE pass over a 2B batch
    flava_batched_train   True | False            the same in TRAINING (each parameter then receives ONE gradient per step instead of two that autograd adds)
    train_side_stream  True | False               CLIP training step: the text tower's forward (and, through autograd, backward) on a side stream
                                                  (same kernels, bit-identical step; -1.3 ... -4 ms of 54 depending on the box)

    flava_grouped  True | False                   FLAVA inference with both towers wanted: the image and the text encoder layer-locked on one
                                                  stream with grouped LayerNorm / GEMM launches (models/flava/transformer.py::run_two_encoders)
                                                  instead of the text tower on a side stream (default since r04: 25.8-25.9 vs 26.1-26.2 ms at
                                                  B = 128, bit-identical outputs)
    train_attentions  True | False                FLAVA training forwards also return the per-layer attention probabilities (recomputed by the
                                                  inference kernel, detached) like the reference's; False: attentions = None (saves one attention
                                                  launch per layer and the S^2 writes)
    flava_attentions  True | False                FLAVA INFERENCE forwards return the per-layer attention probabilities like the reference's (which asks for
                                                  them unconditionally, models/flava/image_encoder.py:217-222).  False: `attentions = None` -- an opt-out for
                                                  callers that never read them (the pre-training losses do not): the fp32 [B, H, S, S] tensors are 4.7 GB of
                                                  writes per forward at B = 128 and a fifth of the step

Environment (read once): MMAMD_TWO_TOWER, MMAMD_SINGLE_STREAM=1.

Retired in r05 (measured losers of r03 / r04; the measurements stay under profiles/, the code in the history before this round):
`residual = "delta_ln"` (bf16 delta GEMMs + one fused residual-add + LayerNorm launch: profiles/r03_delta_ln_ab.txt), `phases = 2` (two
half-batches on two streams with half the chip's CUs each: profiles/r04_phased_schedule_ab.txt) and the per-stream CU budget under it.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, replace

_TWO_TOWER = ("auto", "grouped", "streams")


@dataclass(frozen=True)
class Schedule:
    two_tower: str = "auto"
    side_stream: bool = True
    flava_batched_passes: bool = True
    flava_batched_train: bool = True
    train_side_stream: bool = True
    train_attentions: bool = True
    flava_grouped: bool = True
    flava_attentions: bool = True

    def __post_init__(self):
        if self.two_tower not in _TWO_TOWER:
            raise ValueError(f"two_tower must be one of {_TWO_TOWER}, got {self.two_tower!r}")


def _from_env() -> Schedule:
    """The environment is advisory: an unknown value warns and falls back to the default instead of failing `import multimodal_amd`."""
    import warnings

    def pick(var, allowed, default, legacy=()):
        v = os.environ.get(var)
        if v is None or v == "":
            return default
        v = dict(legacy).get(v, v)
        if v not in allowed:
            warnings.warn(f"{var}={os.environ[var]!r} is not one of {allowed}: using {default!r}", RuntimeWarning, stacklevel=3)
            return default
        return v

    return Schedule(two_tower=pick("MMAMD_TWO_TOWER", _TWO_TOWER, "auto"), side_stream=os.environ.get("MMAMD_SINGLE_STREAM") != "1")


_current = _from_env()


def get_schedule() -> Schedule:
    return _current


def train_side_stream_now() -> bool:
    """Whether a TRAINING forward may put its second tower on the side stream right now: schedule.train_side_stream, and neither a torch.compile
    trace in progress (eager-mode schedule) nor an initialised process group (DistributedDataParallel keeps its gradient-accumulation hooks on the
    stream it was constructed on: a tower whose backward runs elsewhere pays extra syncs there -- the 'AccumulateGrad node's stream does not
    match' warning -- and a HIP-graph capture of the DDP step can break)."""
    import torch

    if not _current.train_side_stream or torch.compiler.is_compiling():
        return False
    return not (torch.distributed.is_available() and torch.distributed.is_initialized())


def set_schedule(**changes) -> Schedule:
    """Replace fields of the process-wide schedule (tools / tests: A-B runs in one process).  Returns the previous schedule."""
    global _current
    prev = _current
    _current = replace(_current, **changes)
    return prev
