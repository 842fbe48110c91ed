"""Headline forward + loss, eager vs torch.compile(fullgraph=True) (dispatcher-op path), B = 256: is the compiled model as fast?
    python tools/compiled_vs_eager.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from tools.nt_ab_models import timed  # noqa: E402


@torch.no_grad()
def main():
    from multimodal_amd.models.clip import clip_vit_b16
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.synthetic import clip_batch

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = clip_vit_b16().to(dev).eval()
    loss_fn = ContrastiveLossWithTemperature().to(dev)
    images, ids = clip_batch(256)
    images, ids = images.to(dev).to(torch.bfloat16), ids.to(dev)

    def step():
        o = model(images, ids)
        return loss_fn(o.embeddings_a, o.embeddings_b)

    cstep = torch.compile(step, backend="aot_eager", fullgraph=True)
    le, lc = step(), cstep()
    print("loss eager", float(le), "compiled", float(lc), "equal", bool(le == lc))
    for rnd in range(3):
        print(f"eager {timed(step, 10):7.3f} ms   compiled {timed(cstep, 10):7.3f} ms", flush=True)


if __name__ == "__main__":
    main()
