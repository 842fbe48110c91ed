"""Helper of tests/test_gpu_loss_w8.py: ONE rank of an EIGHT-process job on a single MI355X (gloo transport: RCCL ranks cannot share a device).
Runs, through the real module path (packed all-gather of HIP tensors, label offset 256 * rank, [256, 2048] logit blocks, backward of every
BackpropType — GLOBAL's reduce-scatter is an all-reduce + own block under gloo):
  1. ContrastiveLossWithTemperature at cfg 3's real size on this rank's block of tests/golden/loss_dist_w8.npz (the reference's own 8-rank run);
  2. the whole weak-scaling step (CLIP towers + loss, eval) on this rank's 4 pairs of tests/golden/clip_w8_step.npz.
Prints one JSON line."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from tests.golden.make_golden_loss_w8 import B, E, ROW_STEP, SMALL, W, inputs, small_batch  # noqa: E402


def sampled_err(z, key, got):
    g = got.detach().double().cpu().numpy()
    return [float(np.abs(g[::ROW_STEP] - z[key + ".rows"]).max()), float(np.abs(g.sum(1) - z[key + ".rowsum"]).max()),
            float(np.abs(g.sum(0) - z[key + ".colsum"]).max())]


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == W
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multimodal_amd.models.clip import CLIPTextEncoder, CLIPViTEncoder
    from multimodal_amd.models.clip.model import CLIP
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import (ContrastiveLossWithTemperature,
                                                                                  contrastive_loss_with_temperature)
    from multimodal_amd.utils.distributed import BackpropType

    golden = ROOT / "tests" / "golden"
    z = np.load(golden / "loss_dist_w8.npz")
    a_all, b_all = inputs()
    res = {"rank": rank}
    for bt in ("GLOBAL", "LOCAL", "NONE"):
        a = a_all[rank * B:(rank + 1) * B].clone().cuda().requires_grad_(True)
        b = b_all[rank * B:(rank + 1) * B].clone().cuda().requires_grad_(True)
        s = torch.nn.Parameter(torch.tensor(float(np.log(1 / 0.07)), dtype=torch.float32, device="cuda"))
        o = contrastive_loss_with_temperature(a, b, s, backprop_type=getattr(BackpropType, bt))
        o.loss.backward()
        res[bt] = {"d_loss": abs(float(o.loss) - float(z[f"{bt}.r{rank}.loss"])), "d_grad_s": abs(float(s.grad) - float(z[f"{bt}.r{rank}.grad_s"])),
                   "d_grad_a": sampled_err(z, f"{bt}.r{rank}.grad_a", a.grad), "d_grad_b": sampled_err(z, f"{bt}.r{rank}.grad_b", b.grad)}
        if bt == "GLOBAL":
            assert tuple(o.logits_a.shape) == (B, W * B)
            res["d_logits_a"] = sampled_err(z, f"r{rank}.logits_a", o.logits_a)
            res["d_logits_b"] = sampled_err(z, f"r{rank}.logits_b", o.logits_b)
            res["d_loss_a"] = abs(float(o.loss_a) - float(z[f"r{rank}.loss_a"]))
    with torch.no_grad():  # the inference path of the module (strided halves of one packed block when they come from CLIP.forward)
        loss_fn = ContrastiveLossWithTemperature().cuda()
        res["fwd_only"] = abs(float(loss_fn(a_all[rank * B:(rank + 1) * B].cuda(), b_all[rank * B:(rank + 1) * B].cuda())) - float(z[f"GLOBAL.r{rank}.loss"]))

    # ---- the whole step on a small CLIP (reference: 8 gloo ranks on CPU, fp32) ------------------------------------------------------------
    zs = np.load(golden / "clip_w8_step.npz")
    from tests._util import assert_checksums

    torch.manual_seed(0)
    vit = CLIPViTEncoder(embedding_dim=SMALL["emb"], heads=SMALL["heads"], layers=SMALL["layers"], patch_size=16, image_size=SMALL["image_size"],
                         width=SMALL["width"])
    txt = CLIPTextEncoder(embedding_dim=SMALL["emb"], context_length=SMALL["ctx"], vocab_size=SMALL["vocab"], width=SMALL["width"],
                          heads=SMALL["heads"], layers=SMALL["layers"])
    model = CLIP(vit, txt)
    assert_checksums(model, zs)
    model = model.cuda().eval()
    images, ids = small_batch()
    Bs = SMALL["B"]
    with torch.no_grad():
        o = model(images[rank * Bs:(rank + 1) * Bs].cuda(), ids[rank * Bs:(rank + 1) * Bs].cuda())
        loss = loss_fn(o.embeddings_a, o.embeddings_b)
    res["step"] = {"loss": float(loss), "ref_loss": float(zs[f"r{rank}.loss"]),
                   "d_emb_a": float(np.abs(o.embeddings_a.float().cpu().numpy() - zs[f"r{rank}.emb_a"]).max()),
                   "d_emb_b": float(np.abs(o.embeddings_b.float().cpu().numpy() - zs[f"r{rank}.emb_b"]).max())}
    mine = loss.reshape(1).float().cpu()
    dist.all_reduce(mine)
    res["step"]["mean_over_ranks"] = float(mine) / world
    res["step"]["ref_one_process"] = float(zs["one_process_loss"])
    dist.barrier()
    print("EIGHT_RANK_RESULT " + json.dumps(res), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
