"""The oracle's contrastive-loss backward pinned to gradients the reference's autograd produced (tests/golden/
make_golden_loss_grad.py): the reference's own gradient KAT (loss 3.8848; grad means 0.0979 / -1.8151 / 3.6792), masked /
label-smoothed / sum-reduced cases with non-trivial upstream weights, and gloo world-2 GLOBAL / LOCAL / NONE per-rank gradients."""
import numpy as np

from oracle import clip_oracle as oc


def test_reference_gradient_kat(golden):
    z = golden("loss_grad.npz")
    ia = z["kat.image_tensor"] @ z["kat.iw"].T + z["kat.ib"]
    tb = z["kat.text_tensor"] @ z["kat.tw"].T + z["kat.tb"]
    s = np.log(1 / 0.07)
    assert abs(float(oc.contrastive_loss_with_temperature(ia, tb, s)["loss"]) - 3.8848) <= 1e-3
    g = oc.contrastive_loss_backward(ia, tb, s)
    ga, gb = g["grad_a"] + g["grad_a_all"], g["grad_b"] + g["grad_b_all"]  # single rank: the gathered rows ARE the local rows
    assert np.abs(ga - z["kat.grad_emb_a"]).max() <= 1e-5 and np.abs(gb - z["kat.grad_emb_b"]).max() <= 1e-5
    # chain rule into the reference's Linear encoders reproduces the numbers its test asserts
    assert abs((ga.T @ z["kat.image_tensor"]).mean() - 0.0979) <= 1e-3
    assert abs(gb.sum(0).mean() - (-1.8151)) <= 1e-3
    assert abs(g["grad_logit_scale"] - 3.6792) <= 1e-3 and abs(g["grad_logit_scale"] - float(z["kat.grad_logit_scale"])) <= 1e-4


def test_single_rank_cases(golden):
    z = golden("loss_grad.npz")
    for name, kw in (("plain", {}), ("smooth_mask", {"label_smoothing": 0.1}), ("sum", {"reduction": "sum"})):
        mask = z[f"{name}.mask"] if f"{name}.mask" in z.files else None
        g = oc.contrastive_loss_backward(z[f"{name}.a"], z[f"{name}.b"], 2.3, mask=mask, grad_out3=(1.7, 0.3, 0.0), **kw)
        assert np.abs(g["grad_a"] + g["grad_a_all"] - z[f"{name}.grad_a"]).max() <= 2e-5, name
        assert np.abs(g["grad_b"] + g["grad_b_all"] - z[f"{name}.grad_b"]).max() <= 2e-5, name
        assert abs(g["grad_logit_scale"] - float(z[f"{name}.grad_s"])) <= 2e-4 * max(1.0, abs(float(z[f"{name}.grad_s"]))), name


def test_two_rank_backprop_types(golden):
    z = golden("loss_grad.npz")
    a_all, b_all = z["dist.a_all"], z["dist.b_all"]
    B = a_all.shape[0] // 2
    s = np.log(1 / 0.07)
    per_rank = [oc.contrastive_loss_backward(a_all[r * B:(r + 1) * B], b_all[r * B:(r + 1) * B], s, a_all, b_all, rank=r) for r in range(2)]
    for r in range(2):
        blk = slice(r * B, (r + 1) * B)
        want = {"GLOBAL": (sum(g["grad_a_all"][blk] for g in per_rank), sum(g["grad_b_all"][blk] for g in per_rank)),
                "LOCAL": (per_rank[r]["grad_a_all"][blk], per_rank[r]["grad_b_all"][blk]), "NONE": (0.0, 0.0)}
        for bt, (xa, xb) in want.items():
            assert np.abs(per_rank[r]["grad_a"] + xa - z[f"dist.{bt}.r{r}.grad_a"]).max() <= 2e-5, (bt, r)
            assert np.abs(per_rank[r]["grad_b"] + xb - z[f"dist.{bt}.r{r}.grad_b"]).max() <= 2e-5, (bt, r)
            assert abs(per_rank[r]["grad_logit_scale"] - float(z[f"dist.{bt}.r{r}.grad_s"])) <= 1e-4, (bt, r)
