"""CLIP ViT-B/16 training step (B = 256): the stack's Linear weights packed (bf16 copy + bf16 transpose) by ONE launch per 64 weights vs one convert and
one transpose launch per weight.  Same-process alternating A/B.   python tools/train_pack_ab.py"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    from multimodal_amd import _autograd
    from multimodal_amd.models.clip import clip_vit_b16
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.synthetic import clip_batch

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = clip_vit_b16().to(dev).train()
    loss_fn = ContrastiveLossWithTemperature().to(dev)
    opt = torch.optim.SGD(list(model.parameters()) + list(loss_fn.parameters()), lr=1e-4)
    images, ids = clip_batch(256)
    images, ids = images.to(dev), ids.to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        o = model(images, ids)
        loss = loss_fn(o.embeddings_a, o.embeddings_b)
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    for rnd in range(3):
        for flag in (False, True):
            _autograd._PACK_WEIGHTS = flag
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            print("packed in one launch " if flag else "per-tensor launches  ", round((time.perf_counter() - t0) / 4 * 1e3, 2), "ms", flush=True)
    _autograd._PACK_WEIGHTS = True


if __name__ == "__main__":
    main()
