"""The global contrastive loss at cfg 3's REAL size from the REFERENCE:  python -m tests.golden.make_golden_loss_w8
  loss_dist_w8.npz   8 gloo ranks (mp.spawn, one thread each), B = 256 per rank, E = 768 (CLIP ViT-L/14's embedding width): every rank runs the
                     reference's contrastive_loss_with_temperature unchanged (contrastive_loss_with_temperature.py:26-47,50-115: two gather_tensor
                     calls, labels 256 * rank + i, [256, 2048] logit blocks) for BackpropType GLOBAL / LOCAL / NONE and backpropagates.
                     Inputs are NOT stored (12.6 MB): the tests regenerate them with `inputs()` below (torch CPU generator, the same image on the
                     GPU box) and check them against the stored float64 checksums first.  Per rank the fixture keeps: the loss, d loss / d logit_scale,
                     and of each [256, 2048] logit block / [256, 768] gradient block every 16th row in full plus float64 row sums and column sums
                     over ALL rows (so every element is covered by two checksums).
  clip_w8_step.npz   the whole weak-scaling step at W = 8 on a small CLIP (the reference's CLIPViTEncoder / CLIPTextEncoder / CLIP under
                     torch.manual_seed(0), B = 4 per rank): per-rank loss of the reference's 8-process gloo run, the embeddings, the one-process
                     loss on the concatenated batch, and weight checksums (the seeded re-creation in multimodal_amd must reproduce them).
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

OUT = Path(__file__).resolve().parent
warnings.filterwarnings("ignore")

W, B, E, SEED, ROW_STEP = 8, 256, 768, 20260930, 16
SMALL = dict(B=4, image_size=64, ctx=16, vocab=512, width=128, emb=128, heads=2, layers=2, seed=11)


def inputs():
    """[W*B, E] image / text features of all ranks: unit rows, text = normalised(image + 0.1 N(0,1) noise): cosine of a matching pair about 0.34 so that the matching pair dominates its row
    (a wrong label offset moves the loss by whole units, not by round-off)."""
    g = torch.Generator().manual_seed(SEED)
    a = torch.nn.functional.normalize(torch.randn(W * B, E, generator=g), dim=1)
    b = torch.nn.functional.normalize(a + 0.1 * torch.randn(W * B, E, generator=g), dim=1)
    return a, b


def checksums(t):
    d = t.double()
    return np.array([float(d.sum()), float(d.abs().sum()), float((d * torch.arange(1, d.numel() + 1, dtype=torch.float64).reshape(d.shape) % 7).sum())])


def small_batch():
    g = torch.Generator().manual_seed(SMALL["seed"])
    n = W * SMALL["B"]
    images = torch.randn(n, 3, SMALL["image_size"], SMALL["image_size"], generator=g)
    ids = torch.randint(1, SMALL["vocab"] - 2, (n, SMALL["ctx"]), generator=g)
    eot = torch.randint(1, SMALL["ctx"], (n,), generator=g)
    ids[torch.arange(n), eot] = SMALL["vocab"] - 1
    return images, ids


def sampled(t):
    d = t.detach().double()
    return t.detach()[::ROW_STEP].numpy().copy(), d.sum(1).numpy(), d.sum(0).numpy()


def _loss_worker(rank, sync, q):
    torch.set_num_threads(1)
    _ref_shim.install()
    import torch.distributed as dist
    from torch import nn
    from torchmultimodal.modules.losses.contrastive_loss_with_temperature import contrastive_loss_with_temperature
    from torchmultimodal.utils.distributed import BackpropType

    dist.init_process_group("gloo", init_method=f"file://{sync}", rank=rank, world_size=W)
    a_all, b_all = inputs()
    out = {}
    for bt in ("GLOBAL", "LOCAL", "NONE"):
        a = a_all[rank * B:(rank + 1) * B].clone().requires_grad_(True)
        b = b_all[rank * B:(rank + 1) * B].clone().requires_grad_(True)
        s = nn.Parameter(torch.tensor(float(np.log(1 / 0.07)), dtype=torch.float32))
        o = contrastive_loss_with_temperature(a, b, s, backprop_type=getattr(BackpropType, bt))
        o.loss.backward()
        p = f"{bt}.r{rank}."
        out[p + "loss"] = np.float32(float(o.loss))
        out[p + "grad_s"] = np.float32(float(s.grad))
        for nm, t in (("grad_a", a.grad), ("grad_b", b.grad)):
            out[p + nm + ".rows"], out[p + nm + ".rowsum"], out[p + nm + ".colsum"] = sampled(t)
        if bt == "GLOBAL":
            out[f"r{rank}.loss_a"], out[f"r{rank}.loss_b"] = np.float32(float(o.loss_a)), np.float32(float(o.loss_b))
            for nm, t in (("logits_a", o.logits_a), ("logits_b", o.logits_b)):
                assert t.shape == (B, W * B)
                out[f"r{rank}.{nm}.rows"], out[f"r{rank}.{nm}.rowsum"], out[f"r{rank}.{nm}.colsum"] = sampled(t)
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def _small_clip_reference():
    from torchmultimodal.models.clip.image_encoder import CLIPViTEncoder
    from torchmultimodal.models.clip.model import CLIP
    from torchmultimodal.models.clip.text_encoder import CLIPTextEncoder

    torch.manual_seed(0)
    vit = CLIPViTEncoder(embedding_dim=SMALL["emb"], heads=SMALL["heads"], layers=SMALL["layers"], patch_size=16, image_size=SMALL["image_size"],
                         width=SMALL["width"])
    txt = CLIPTextEncoder(embedding_dim=SMALL["emb"], context_length=SMALL["ctx"], vocab_size=SMALL["vocab"], width=SMALL["width"],
                          heads=SMALL["heads"], layers=SMALL["layers"])
    return CLIP(vit, txt).eval()


def _step_worker(rank, sync, q):
    torch.set_num_threads(1)
    _ref_shim.install()
    import torch.distributed as dist
    from torchmultimodal.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature

    dist.init_process_group("gloo", init_method=f"file://{sync}", rank=rank, world_size=W)
    model, loss_fn = _small_clip_reference(), ContrastiveLossWithTemperature()
    images, ids = small_batch()
    Bs = SMALL["B"]
    with torch.no_grad():
        o = model(images[rank * Bs:(rank + 1) * Bs], ids[rank * Bs:(rank + 1) * Bs])
        loss = loss_fn(o.embeddings_a, o.embeddings_b)
    q.put({f"r{rank}.loss": np.float32(float(loss)), f"r{rank}.emb_a": o.embeddings_a.numpy().copy(), f"r{rank}.emb_b": o.embeddings_b.numpy().copy()})
    dist.barrier()
    dist.destroy_process_group()


def _spawn(worker):
    ctx = mp.get_context("spawn")
    st = {}
    with tempfile.TemporaryDirectory() as d:
        q = ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, os.path.join(d, "sync"), q)) for r in range(W)]
        [p.start() for p in procs]
        for _ in range(W):
            st.update(q.get())
        [p.join() for p in procs]
    return st


def main():
    _ref_shim.install()
    a_all, b_all = inputs()
    st = {"meta": np.array([W, B, E, SEED, ROW_STEP], dtype=np.int64), "a_all.checksums": checksums(a_all), "b_all.checksums": checksums(b_all)}
    st.update(_spawn(_loss_worker))
    # single-process cross-check of the 8-rank run (SURVEY 8c protocol (3)): mean of the rank losses == the loss on the concatenated batch
    from torchmultimodal.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature

    with torch.no_grad():
        one = float(ContrastiveLossWithTemperature()(a_all, b_all))
    mean = float(np.mean([st[f"GLOBAL.r{r}.loss"] for r in range(W)]))
    print(f"W=8 mean of rank losses {mean:.6f}  one-process loss on [2048, 768] {one:.6f}")
    assert abs(mean - one) < 2e-5
    st["one_process_loss"] = np.float32(one)
    np.savez_compressed(OUT / "loss_dist_w8.npz", **st)
    print("loss_dist_w8.npz:", len(st), "arrays,", (OUT / "loss_dist_w8.npz").stat().st_size // 1024, "KiB")

    step = {"meta": np.array([W, SMALL["B"]], dtype=np.int64)}
    step.update(_spawn(_step_worker))
    model = _small_clip_reference()
    images, ids = small_batch()
    with torch.no_grad():
        o = model(images, ids)
        step["one_process_loss"] = np.float32(float(ContrastiveLossWithTemperature()(o.embeddings_a, o.embeddings_b)))
    sd = model.state_dict()
    step["keys"] = np.array(list(sd.keys()))
    step["sums"] = np.array([float(v.double().sum()) for v in sd.values()])
    step["asums"] = np.array([float(v.double().abs().sum()) for v in sd.values()])
    mean = float(np.mean([step[f"r{r}.loss"] for r in range(W)]))
    print(f"small CLIP step: mean of rank losses {mean:.6f}  one-process {float(step['one_process_loss']):.6f}")
    assert abs(mean - float(step["one_process_loss"])) < 2e-5
    np.savez_compressed(OUT / "clip_w8_step.npz", **step)
    print("clip_w8_step.npz:", len(step), "arrays")


if __name__ == "__main__":
    main()
