"""Which host-side call sites issue device-to-device copies (aten::copy_ / clone / contiguous) in one CLIP training step -- or, with --infer, in one
step of the headline forward + loss?     python tools/train_copy_trace.py [--infer]"""
import sys
from collections import Counter
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    from multimodal_amd.models.clip import clip_vit_b16
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.synthetic import clip_batch

    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    infer = "--infer" in sys.argv
    model = clip_vit_b16().to(dev)
    model = model.eval() if infer else model.train()
    loss_fn = ContrastiveLossWithTemperature().to(dev)
    opt = torch.optim.SGD(list(model.parameters()) + list(loss_fn.parameters()), lr=1e-4)
    images, ids = clip_batch(256 if infer else 64)
    images, ids = images.to(dev), ids.to(dev)

    def step():
        if infer:
            with torch.no_grad():
                out = model(images, ids)
                return loss_fn(out.embeddings_a, out.embeddings_b)
        opt.zero_grad(set_to_none=True)
        out = model(images, ids)
        loss = loss_fn(out.embeddings_a, out.embeddings_b)
        loss.backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    c = Counter()
    for ev in prof.events():
        if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::zeros", "aten::zero_", "aten::fill_", "aten::cat", "aten::to", "aten::_to_copy"):
            frames = [f for f in (ev.stack or []) if "multimodal_amd" in f or "tools/" in f or "optim" in f or "autograd" in f]
            key = (ev.name, str(ev.input_shapes)[:60], (frames[0] if frames else "?")[-90:])
            c[key] += 1
    for (name, shp, fr), n in c.most_common(40):
        print(f"{n:4d} {name:18s} {shp:60s} {fr}")


if __name__ == "__main__":
    main()
