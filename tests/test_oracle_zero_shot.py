"""Zero-shot / retrieval read-outs, CPU suite: the numpy oracle against the torch expressions of the
reference's examples (fixture tests/golden/zero_shot.npz, make_golden_zero_shot.py)."""
import numpy as np

from oracle import clip_oracle as oc


def test_oracle_zero_shot_matches_reference_expressions(golden):
    z = golden("zero_shot.npz")
    w = np.stack([oc.zero_shot_class_embedding(p, np.float32) for p in z["prompts"]], axis=1)
    np.testing.assert_allclose(w, z["classifier"], atol=2e-7)
    np.testing.assert_allclose(oc.zero_shot_logits(z["feats"], z["classifier"], 100.0, np.float32), z["logits"], atol=2e-5)
    assert oc.topk_hits(z["logits"], z["target"], (1, 5, 10)) == z["acc"].tolist()
    got = [oc.recall_at_k(s, k) for s in (z["sim"], z["sim"].T) for k in (1, 5)]
    np.testing.assert_allclose(got, z["recall"], atol=1e-7)
    assert 0.0 < z["recall"].min() and z["recall"][0] < z["recall"][1] <= 1.0  # a fixture with both hits and misses
    assert 0 < z["acc"][0] < z["acc"][2] < len(z["target"])
