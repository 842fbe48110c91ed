"""Gradient fixtures of the contrastive loss from the REFERENCE:  python -m tests.golden.make_golden_loss_grad
  loss_grad.npz  the reference's own gradient KAT setting (tests/modules/losses/test_contrastive_loss_with_temperature.py:129-199:
                 seed 0, image Linear(8,3), text Linear(5,3), global batch 4 -> loss 3.8848, grad means 0.0979 / -1.8151 / 3.6792),
                 run on CPU (world 1) with the full gradient tensors, plus larger random cases: masks, label smoothing, sum
                 reduction, and gloo world 2 / GLOBAL, LOCAL, NONE per-rank gradients of the embeddings and logit_scale.
"""
from __future__ import annotations

import os
import sys
import tempfile
import warnings
from pathlib import Path

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.golden import _ref_shim  # noqa: E402
from tests.golden.make_golden import seed  # noqa: E402

OUT = Path(__file__).resolve().parent
warnings.filterwarnings("ignore")


def tnp(t):
    return t.detach().numpy().copy()


def _worker(rank, world, sync, a_all, b_all, bt_name, q):
    _ref_shim.install()
    import torch.distributed as dist
    from torchmultimodal.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from torchmultimodal.utils.distributed import BackpropType

    dist.init_process_group("gloo", init_method=f"file://{sync}", rank=rank, world_size=world)
    B = a_all.shape[0] // world
    a = a_all[rank * B:(rank + 1) * B].clone().requires_grad_(True)
    b = b_all[rank * B:(rank + 1) * B].clone().requires_grad_(True)
    loss_fn = ContrastiveLossWithTemperature()
    loss = loss_fn(a, b, backprop_type=getattr(BackpropType, bt_name))
    loss.backward()
    q.put((rank, float(loss), a.grad.numpy().copy(), b.grad.numpy().copy(), float(loss_fn.logit_scale.grad)))
    dist.barrier()
    dist.destroy_process_group()


def main():
    _ref_shim.install()
    from torch import nn
    from torchmultimodal.modules.losses.contrastive_loss_with_temperature import (ContrastiveLossWithTemperature,
                                                                                  contrastive_loss_with_temperature)

    st = {}
    # ---- the reference's KAT (fixture instantiation order = test argument order: image_tensor, text_tensor, image_encoder, text_encoder)
    seed(0)
    image_tensor = torch.randn(4, 8)
    text_tensor = torch.randn(4, 5)
    image_encoder = nn.Linear(8, 3)
    text_encoder = nn.Linear(5, 3)
    loss_fn = ContrastiveLossWithTemperature()
    ia, tb = image_encoder(image_tensor), text_encoder(text_tensor)
    ia.retain_grad(); tb.retain_grad()
    loss = loss_fn(ia, tb)
    loss.backward()
    kat = (float(loss), float(image_encoder.weight.grad.mean()), float(text_encoder.bias.grad.mean()), float(loss_fn.logit_scale.grad))
    print("KAT:", kat)
    assert abs(kat[0] - 3.8848) < 1e-3 and abs(kat[1] - 0.0979) < 1e-3 and abs(kat[2] + 1.8151) < 1e-3 and abs(kat[3] - 3.6792) < 1e-3
    st.update({"kat.image_tensor": tnp(image_tensor), "kat.text_tensor": tnp(text_tensor), "kat.iw": tnp(image_encoder.weight),
               "kat.ib": tnp(image_encoder.bias), "kat.tw": tnp(text_encoder.weight), "kat.tb": tnp(text_encoder.bias),
               "kat.loss": np.float32(kat[0]), "kat.grad_emb_a": tnp(ia.grad), "kat.grad_emb_b": tnp(tb.grad),
               "kat.grad_iw": tnp(image_encoder.weight.grad), "kat.grad_tbias": tnp(text_encoder.bias.grad),
               "kat.grad_logit_scale": np.float32(kat[3])})
    # ---- larger single-rank cases through the functional
    g = torch.Generator().manual_seed(3)
    for name, B, E, kw, use_mask in (("plain", 37, 64, {}, False), ("smooth_mask", 50, 96, {"label_smoothing": 0.1}, True),
                                     ("sum", 16, 30, {"reduction": "sum"}, False)):
        a = torch.nn.functional.normalize(torch.randn(B, E, generator=g), dim=1).requires_grad_(True)
        b = torch.nn.functional.normalize(torch.randn(B, E, generator=g), dim=1).requires_grad_(True)
        s = nn.Parameter(torch.tensor(2.3))
        mask = (torch.rand(B, generator=g) > 0.3) if use_mask else None
        o = contrastive_loss_with_temperature(a, b, s, mask=mask, cross_entropy_kwargs=kw or None)
        (o.loss * 1.7 + 0.3 * o.loss_a).backward()  # exercises all three upstream weights
        st.update({f"{name}.a": tnp(a), f"{name}.b": tnp(b), f"{name}.loss": tnp(o.loss), f"{name}.grad_a": tnp(a.grad),
                   f"{name}.grad_b": tnp(b.grad), f"{name}.grad_s": tnp(s.grad)})
        if mask is not None:
            st[f"{name}.mask"] = tnp(mask)
    # ---- gloo world 2: per-rank gradients for the three backprop types
    a_all = torch.nn.functional.normalize(torch.randn(12, 16, generator=g), dim=1)
    b_all = torch.nn.functional.normalize(torch.randn(12, 16, generator=g), dim=1)
    st["dist.a_all"], st["dist.b_all"] = tnp(a_all), tnp(b_all)
    ctx = mp.get_context("spawn")
    for bt in ("GLOBAL", "LOCAL", "NONE"):
        with tempfile.TemporaryDirectory() as d:
            q = ctx.Queue()
            procs = [ctx.Process(target=_worker, args=(r, 2, os.path.join(d, "sync"), a_all.clone(), b_all.clone(), bt, q)) for r in range(2)]
            [p.start() for p in procs]
            res = sorted(q.get() for _ in range(2))
            [p.join() for p in procs]
        for r, loss, ga, gb, gs in res:
            st.update({f"dist.{bt}.r{r}.loss": np.float32(loss), f"dist.{bt}.r{r}.grad_a": ga, f"dist.{bt}.r{r}.grad_b": gb,
                       f"dist.{bt}.r{r}.grad_s": np.float32(gs)})
    np.savez_compressed(OUT / "loss_grad.npz", **st)
    print("written", len(st), "arrays")


if __name__ == "__main__":
    main()
