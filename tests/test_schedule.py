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
    assert (s.two_tower, s.residual, s.side_stream, s.flava_batched_passes, s.train_side_stream) == ("auto", "epilogue", True, True, True)
    with pytest.raises(ValueError):
        Schedule(two_tower="both")
    with pytest.raises(ValueError):
        Schedule(residual="bf16")
    with pytest.raises(Exception):
        s.two_tower = "streams"  # frozen
    prev = set_schedule(two_tower="streams", train_side_stream=False)
    try:
        assert get_schedule().two_tower == "streams" and get_schedule().train_side_stream is False
        assert prev.two_tower in ("auto", "grouped", "streams")
        with pytest.raises(ValueError):
            set_schedule(residual="nope")
        assert get_schedule().two_tower == "streams"  # a rejected change leaves the record as it was
    finally:
        set_schedule(two_tower=prev.two_tower, train_side_stream=prev.train_side_stream)
    assert get_schedule() == prev


def test_environment_is_read_once_at_import():
    code = "from multimodal_amd.schedule import get_schedule as g; s = g(); print(s.two_tower, s.residual, s.side_stream)"
    env = dict(os.environ, MMAMD_TWO_TOWER="grouped", MMAMD_RESIDUAL="delta_ln", MMAMD_SINGLE_STREAM="1", PYTHONPATH=str(ROOT))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["grouped", "delta_ln", "False"]
    # the environment is advisory: an unknown value warns and falls back to the default, a round-2 spelling is mapped -- the import never fails
    bad = subprocess.run([sys.executable, "-c", code], env=dict(env, MMAMD_TWO_TOWER="sideways", MMAMD_RESIDUAL="fp32"), capture_output=True, text=True, timeout=120)
    assert bad.returncode == 0, bad.stderr[-2000:]
    assert bad.stdout.split() == ["auto", "epilogue", "False"] and "MMAMD_TWO_TOWER" in bad.stderr
    ph = subprocess.run([sys.executable, "-c", "from multimodal_amd.schedule import get_schedule as g; print(g().phases, g().phase_lead)"],
                        env=dict(env, MMAMD_PHASES="2"), capture_output=True, text=True, timeout=120)
    assert ph.returncode == 0 and ph.stdout.split() == ["2", "4"], ph.stderr[-2000:]


def test_phases_fields():
    from multimodal_amd.schedule import Schedule

    assert Schedule().phases == 1
    assert Schedule(phases=2, phase_lead=6).phase_lead == 6
    with pytest.raises(ValueError):
        Schedule(phases=3)
    with pytest.raises(ValueError):
        Schedule(phase_lead=-1)


def test_round4_schedule_fields_and_mask_flags():
    """r04 additions: flava_attentions (default on = the reference's behaviour) and the 2-bit `causal` flag of the general attention kernels."""
    import torch

    from multimodal_amd import ops
    from multimodal_amd.schedule import Schedule

    s = Schedule()
    assert s.flava_attentions is True and s.flava_grouped is True and s.phases == 1 and s.train_attentions is True
    assert ops.AttnMask().causal_flags == 0 and ops.AttnMask(causal=True).causal_flags == 1
    km = torch.ones(2, 5, dtype=torch.uint8)
    assert ops.AttnMask(causal=True, key_mask=km, key_mask_last_row=True).causal_flags == 3
    assert ops.AttnMask(causal=False, key_mask=km, key_mask_last_row=True).causal_flags == 2
    assert ops.AttnMask(causal=True, key_mask_last_row=True).causal_flags == 1  # without a key mask the bit has nothing to bind
    assert ops.AttnMask().empty and not ops.AttnMask(key_mask=km).empty
