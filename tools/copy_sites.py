"""Which Python call sites issue copy-like torch calls (copy_ / clone / contiguous / to / cat / zeros / fill_ / item ...) in one step of the headline forward
+ loss -- by wrapping the torch entry points and recording the caller's frame (the torch profiler's with_stack gave no Python frames in this build).
    python tools/copy_sites.py [--train]"""
import sys
import traceback
from collections import Counter
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

SITES = Counter()
ACTIVE = [False]


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        if ACTIVE[0]:
            fr = [x for x in traceback.extract_stack(limit=12)[:-1] if "multimodal_amd" in x.filename or "tools/" in x.filename or "bench.py" in x.filename]
            t = a[0] if a and isinstance(a[0], torch.Tensor) else None
            where = f"{Path(fr[-1].filename).name}:{fr[-1].lineno} {fr[-1].line[:70]}" if fr else "?"
            SITES[(f"{getattr(owner, '__name__', owner)}.{name}", str(tuple(t.shape)) + str(t.dtype)[6:] + ("/cuda" if t.is_cuda else "/cpu") if t is not None else "", where)] += 1
        return orig(*a, **k)

    setattr(owner, name, f)


def main():
    from multimodal_amd.models.clip import clip_vit_b16
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.synthetic import clip_batch

    train = "--train" in sys.argv
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    model = clip_vit_b16().to(dev)
    model = model.train() if train else model.eval()
    loss_fn = ContrastiveLossWithTemperature().to(dev)
    images, ids = clip_batch(256)
    images, ids = images.to(dev), ids.to(dev)

    def step():
        if train:
            out = model(images, ids)
            loss_fn(out.embeddings_a, out.embeddings_b).backward()
            return
        with torch.no_grad():
            out = model(images, ids)
            return loss_fn(out.embeddings_a, out.embeddings_b)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    for name in ("copy_", "clone", "contiguous", "to", "fill_", "zero_", "item", "float", "long", "int", "type", "cuda", "cpu", "masked_fill_", "clamp_", "clamp", "exp", "argmax"):
        wrap(torch.Tensor, name)
    for name in ("cat", "zeros", "ones", "full", "arange", "tensor", "empty_like", "zeros_like", "stack", "where", "clamp", "exp"):
        wrap(torch, name)
    ACTIVE[0] = True
    step()
    torch.cuda.synchronize()
    ACTIVE[0] = False
    for (fn, shp, where), n in SITES.most_common(60):
        print(f"{n:4d} {fn:22s} {shp:34s} {where}")


if __name__ == "__main__":
    main()
