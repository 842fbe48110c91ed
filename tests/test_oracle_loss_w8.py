"""The oracle's global contrastive loss at cfg 3's real size (W = 8 ranks, B = 256 per rank, E = 768: [256, 768] x [2048, 768]^T logit blocks,
labels 256 * rank + i) pinned to the reference's own 8-process gloo run (tests/golden/make_golden_loss_w8.py ->
loss_dist_w8.npz; reference: modules/losses/contrastive_loss_with_temperature.py:26-47,90-107).  CPU only."""
import numpy as np

from oracle import clip_oracle as oc
from tests.golden.make_golden_loss_w8 import B, E, ROW_STEP, W, checksums, inputs


def regenerated_inputs(z):
    a_all, b_all = inputs()
    assert tuple(z["meta"][:3]) == (W, B, E)
    for t, name in ((a_all, "a_all"), (b_all, "b_all")):
        want, got = z[f"{name}.checksums"], checksums(t)
        assert np.all(np.abs(got - want) <= 1e-9 * np.maximum(1.0, np.abs(want))), f"{name}: the seeded inputs did not regenerate ({got} vs {want})"
    return a_all.numpy(), b_all.numpy()


def check_sampled(z, key, got, tol):
    """`got` [rows, cols] float64 against a block stored as every ROW_STEP-th row + row sums + column sums of all rows."""
    got = np.asarray(got, dtype=np.float64)
    assert np.abs(got[::ROW_STEP] - z[key + ".rows"]).max() <= tol, key
    n_r, n_c = got.shape
    assert np.abs(got.sum(1) - z[key + ".rowsum"]).max() <= tol * n_c ** 0.5 * 4, key
    assert np.abs(got.sum(0) - z[key + ".colsum"]).max() <= tol * n_r ** 0.5 * 4, key


def test_oracle_reproduces_the_reference_eight_rank_run(golden):
    z = golden("loss_dist_w8.npz")
    a_all, b_all = regenerated_inputs(z)
    s = np.log(1 / 0.07)
    per_rank = []
    for r in range(W):
        a, b = a_all[r * B:(r + 1) * B], b_all[r * B:(r + 1) * B]
        f = oc.contrastive_loss_with_temperature(a, b, s, a_all, b_all, rank=r, dtype=np.float64)
        assert f["logits_a"].shape == (B, W * B)
        assert abs(float(f["loss"]) - float(z[f"GLOBAL.r{r}.loss"])) <= 2e-5
        assert abs(float(f["loss_a"]) - float(z[f"r{r}.loss_a"])) <= 2e-5 and abs(float(f["loss_b"]) - float(z[f"r{r}.loss_b"])) <= 2e-5
        check_sampled(z, f"r{r}.logits_a", f["logits_a"], 1e-5)
        check_sampled(z, f"r{r}.logits_b", f["logits_b"], 1e-5)
        per_rank.append(oc.contrastive_loss_backward(a, b, s, a_all, b_all, rank=r))
    assert abs(np.mean([float(z[f"GLOBAL.r{r}.loss"]) for r in range(W)]) - float(z["one_process_loss"])) <= 2e-5
    for r in range(W):
        blk = slice(r * B, (r + 1) * B)
        extra = {"GLOBAL": (sum(g["grad_a_all"][blk] for g in per_rank), sum(g["grad_b_all"][blk] for g in per_rank)),
                 "LOCAL": (per_rank[r]["grad_a_all"][blk], per_rank[r]["grad_b_all"][blk]), "NONE": (0.0, 0.0)}
        for bt, (xa, xb) in extra.items():
            check_sampled(z, f"{bt}.r{r}.grad_a", per_rank[r]["grad_a"] + xa, 2e-7)
            check_sampled(z, f"{bt}.r{r}.grad_b", per_rank[r]["grad_b"] + xb, 2e-7)
            assert abs(per_rank[r]["grad_logit_scale"] - float(z[f"{bt}.r{r}.grad_s"])) <= 1e-5, (bt, r)
