"""Helper of tests/test_gpu_two_ranks.py: ONE rank of a two-process job on a single MI355X (gloo transport: two RCCL ranks cannot share a
device).  Runs ContrastiveLossWithTemperature on this rank's block of the reference fixture (tests/golden/loss_grad.npz, `dist.*`) through the
real module path — packed all-gather, label offsets B * rank, backward for every BackpropType (GLOBAL's reduce-scatter is an all-reduce +
own block under gloo) — and prints one JSON line."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def _small_clip():
    from multimodal_amd.models.clip import CLIPTextEncoder, CLIPViTEncoder
    from multimodal_amd.models.clip.model import CLIP

    torch.manual_seed(0)
    vit = CLIPViTEncoder(embedding_dim=128, heads=2, layers=2, patch_size=16, image_size=64, width=128)
    txt = CLIPTextEncoder(embedding_dim=128, context_length=16, vocab_size=512, width=128, heads=2, layers=2)
    return CLIP(vit, txt).cuda().eval()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature as _Loss

    # the whole step, first WITHOUT a process group on the concatenated batch of both ranks: the value the W = 2 job must reproduce
    g = torch.Generator().manual_seed(7)
    images = torch.randn(16, 3, 64, 64, generator=g).cuda()
    ids = torch.randint(1, 500, (16, 16), generator=g)
    ids[:, -1] = 511
    ids = ids.cuda()
    model, loss_1p = _small_clip(), _Loss().cuda()
    with torch.no_grad():
        o = model(images, ids)
        single = float(loss_1p(o.embeddings_a, o.embeddings_b))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.distributed import BackpropType

    z = np.load(Path(__file__).parent / "golden" / "loss_grad.npz")
    a_all, b_all = torch.from_numpy(z["dist.a_all"]), torch.from_numpy(z["dist.b_all"])
    B = a_all.shape[0] // world
    res = {"rank": rank}
    loss_fn = ContrastiveLossWithTemperature().cuda()
    for bt in ("GLOBAL", "LOCAL", "NONE"):
        a = a_all[rank * B:(rank + 1) * B].clone().cuda().requires_grad_(True)
        b = b_all[rank * B:(rank + 1) * B].clone().cuda().requires_grad_(True)
        loss_fn.zero_grad()
        loss = loss_fn(a, b, backprop_type=getattr(BackpropType, bt))
        loss.backward()
        res[bt] = {"loss": float(loss), "d_loss": abs(float(loss) - float(z[f"dist.GLOBAL.r{rank}.loss"])),
                   "d_grad_a": float(np.abs(a.grad.cpu().numpy() - z[f"dist.{bt}.r{rank}.grad_a"]).max()),
                   "d_grad_b": float(np.abs(b.grad.cpu().numpy() - z[f"dist.{bt}.r{rank}.grad_b"]).max()),
                   "d_grad_s": abs(float(loss_fn.logit_scale.grad) - float(z[f"dist.{bt}.r{rank}.grad_s"]))}
    with torch.no_grad():  # the forward-only (inference) path: column views of one packed block, gathered without packing copies
        a = a_all[rank * B:(rank + 1) * B].cuda()
        b = b_all[rank * B:(rank + 1) * B].cuda()
        res["fwd_only"] = abs(float(loss_fn(a, b)) - float(z[f"dist.GLOBAL.r{rank}.loss"]))
    with torch.no_grad():  # the weak-scaling step of bench.py at W = 2: this rank's half of the batch, global negatives
        Bh = 16 // world
        o = model(images[rank * Bh:(rank + 1) * Bh], ids[rank * Bh:(rank + 1) * Bh])
        mine = loss_1p(o.embeddings_a, o.embeddings_b).reshape(1).cpu()
    dist.all_reduce(mine)
    res["step_loss_mean_over_ranks"] = float(mine) / world
    res["step_loss_single_process"] = single
    dist.barrier()
    print("TWO_RANK_RESULT " + json.dumps(res), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
