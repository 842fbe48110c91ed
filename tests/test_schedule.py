"""The schedule object (multimodal_amd/schedule.py): one frozen settings record, validated, read from the environment once, changed only through
set_schedule (which returns the previous record).  CPU-only."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_defaults_and_validation():
    from multimodal_amd.schedule import Schedule, get_schedule, set_schedule

    s = Schedule()
    assert (s.two_tower, s.side_stream, s.flava_batched_passes, s.train_side_stream) == ("auto", True, True, True)
    with pytest.raises(ValueError):
        Schedule(two_tower="both")
    with pytest.raises(Exception):
        s.two_tower = "streams"  # frozen
    prev = set_schedule(two_tower="streams", train_side_stream=False)
    try:
        assert get_schedule().two_tower == "streams" and get_schedule().train_side_stream is False
        assert prev.two_tower in ("auto", "grouped", "streams")
        with pytest.raises(ValueError):
            set_schedule(two_tower="nope")
        assert get_schedule().two_tower == "streams"  # a rejected change leaves the record as it was
    finally:
        set_schedule(two_tower=prev.two_tower, train_side_stream=prev.train_side_stream)
    assert get_schedule() == prev


def test_environment_is_read_once_at_import():
    code = "from multimodal_amd.schedule import get_schedule as g; s = g(); print(s.two_tower, s.side_stream)"
    env = dict(os.environ, MMAMD_TWO_TOWER="grouped", MMAMD_SINGLE_STREAM="1", PYTHONPATH=str(ROOT))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["grouped", "False"]
    # the environment is advisory: an unknown value warns and falls back to the default -- the import never fails; the knobs retired in
    # r05 (MMAMD_RESIDUAL, MMAMD_PHASES) are simply not read any more
    bad = subprocess.run([sys.executable, "-c", code], env=dict(env, MMAMD_TWO_TOWER="sideways", MMAMD_RESIDUAL="delta_ln", MMAMD_PHASES="2"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode == 0, bad.stderr[-2000:]
    assert bad.stdout.split() == ["auto", "False"] and "MMAMD_TWO_TOWER" in bad.stderr


def test_retired_schedule_fields_are_gone():
    """r05 hygiene: the measured losers `residual = "delta_ln"`, `phases = 2` / `phase_lead` are no longer part of the record."""
    from multimodal_amd.schedule import Schedule

    for kw in ({"phases": 2}, {"phase_lead": 4}, {"residual": "delta_ln"}):
        with pytest.raises(TypeError):
            Schedule(**kw)


def test_round4_schedule_fields_and_mask_flags():
    """r04 additions: flava_attentions (default on = the reference's behaviour) and the 2-bit `causal` flag of the general attention kernels."""
    import torch

    from multimodal_amd import ops
    from multimodal_amd.schedule import Schedule

    s = Schedule()
    assert s.flava_attentions is True and s.flava_grouped is True and s.train_attentions is True
    assert ops.AttnMask().causal_flags == 0 and ops.AttnMask(causal=True).causal_flags == 1
    km = torch.ones(2, 5, dtype=torch.uint8)
    assert ops.AttnMask(causal=True, key_mask=km, key_mask_last_row=True).causal_flags == 3
    assert ops.AttnMask(causal=False, key_mask=km, key_mask_last_row=True).causal_flags == 2
    assert ops.AttnMask(causal=True, key_mask_last_row=True).causal_flags == 1  # without a key mask the bit has nothing to bind
    assert ops.AttnMask().empty and not ops.AttnMask(key_mask=km).empty
