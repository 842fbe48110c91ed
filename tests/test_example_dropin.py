"""Example-level drop-in (north_star: "drops into the existing examples"): the reference's own example modules —
examples/flava/native/model.py:39-49 (FLAVAPreTrainModule -> flava_model_for_pretraining) and examples/mugen/retrieval/model.py:45-50
(VideoCLIPLightningModule -> CLIP + ContrastiveLossWithTemperature, via examples/mugen/retrieval/video_clip.py) — are imported UNCHANGED with
`torchmultimodal.*` aliased to `multimodal_amd.*`, constructed, and their state_dicts (names and shapes) compared with what the same example
code builds on the real reference package.  CPU test; needs the reference checkout (skipped on the GPU box)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

from tests.golden import _ref_shim

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.skipif(not _ref_shim.reference_available(), reason="needs the reference checkout (build container only)")


def _probe(mode):
    res = subprocess.run([sys.executable, str(ROOT / "tests" / "_example_dropin_probe.py"), mode, _ref_shim.REFERENCE_ROOT, str(ROOT)],
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("PROBE_JSON ")][-1]
    return json.loads(line[len("PROBE_JSON "):])


def test_reference_examples_construct_on_the_aliased_modules():
    ours, ref = _probe("alias"), _probe("reference")
    # the example classes really wrapped THIS package's modules
    assert ours["mugen_types"] == ["CLIP", "ContrastiveLossWithTemperature", "multimodal_amd"]
    assert ref["mugen_types"][:2] == ["CLIP", "ContrastiveLossWithTemperature"] and ref["mugen_types"][2] == "torchmultimodal"
    assert ours["flava_native_types"] == ref["flava_native_types"] == ["FLAVAForPreTraining", "FLAVAModel", "FLAVAPretrainingLoss", "DalleVAEEncoder"]
    # same parameter / buffer names and shapes: checkpoints written by the example on either package load in the other
    for key in ("flava_native", "mugen_retrieval"):
        assert ours[key].keys() == ref[key].keys(), (key, sorted(set(ours[key]) ^ set(ref[key]))[:10])
        bad = [k for k in ref[key] if ours[key][k] != ref[key][k]]
        assert not bad, (key, bad[:10])
    assert len(ref["flava_native"]) > 500 and len(ref["mugen_retrieval"]) > 100
    assert ours["mugen_logit_scale"] == pytest.approx(ref["mugen_logit_scale"])  # logit_scale = log(1 / 0.07) handling of the loss ctor
