"""Generates tests/golden/zero_shot.npz: the zero-shot / retrieval read-outs of the reference's examples evaluated with torch on
seeded inputs.  examples/flava/native/utils.py and examples/flava/coco_zero_shot.py cannot be imported here (hydra, omegaconf,
torchvision datasets, HF tokenizers at module scope), so the generator evaluates the same torch expressions those functions consist of
(utils.py:108-111,117-123,141-142; coco_zero_shot.py:24-31,84-88) -- topk / eq / sum on the scores, exactly what the product replaces
with mmamd_target_rank.

    python tests/golden/make_golden_zero_shot.py
"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    g = torch.Generator().manual_seed(0)
    out = {}
    C, T, E, N = 37, 7, 96, 53
    prompts = torch.randn(C, T, E, generator=g)
    weights = []
    for c in range(C):
        e = prompts[c] / prompts[c].norm(dim=-1, keepdim=True)
        m = e.mean(dim=0)
        weights.append(m / m.norm())
    classifier = torch.stack(weights, dim=1)                       # [E, C]
    feats = torch.randn(N, E, generator=g) * 3 + classifier.t()[torch.randint(0, C, (N,), generator=g)] * 4
    f = feats / feats.norm(dim=-1, keepdim=True)
    logits = 100.0 * f @ classifier
    target = torch.randint(0, C, (N,), generator=g)
    target[: N // 2] = logits[: N // 2].argmax(1)                   # a mix of hits and misses
    topk = (1, 5, 10)
    pred = logits.topk(max(topk), 1, True, True)[1].t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    acc = [float(correct[:k].reshape(-1).float().sum(0, keepdim=True).numpy()[0]) for k in topk]
    out.update(prompts=prompts.numpy(), classifier=classifier.numpy(), feats=feats.numpy(), logits=logits.numpy(),
               target=target.numpy(), acc=np.asarray(acc))
    # retrieval
    M = 61
    img = torch.randn(M, E, generator=g)
    txt = img + 5.0 * torch.randn(M, E, generator=g)
    a = torch.nn.functional.normalize(img, dim=-1)
    b = torch.nn.functional.normalize(txt, dim=-1)
    sim = a @ b.t()
    rec = []
    for s in (sim, sim.t()):
        for k in (1, 5):
            targets = torch.arange(M).view(M, -1)
            _, idx = torch.topk(s, k)
            rec.append(float(targets.eq(idx).sum() / M))
    out.update(img=img.numpy(), txt=txt.numpy(), sim=sim.numpy(), recall=np.asarray(rec))
    np.savez_compressed(os.path.join(HERE, "zero_shot.npz"), **out)
    print("acc", acc, "recall", rec)


if __name__ == "__main__":
    main()
