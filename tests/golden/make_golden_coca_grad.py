"""CoCa training-step gradients from the REFERENCE (torch autograd):  python -m tests.golden.make_golden_coca_grad
  coca_grad.npz  the small parallel-pooler coca_vit models of make_golden_coca.py (seed 51: 64-wide heads; seed 53: 96-wide pooler
                 heads), CoCaForPretraining in train mode on the same batch: the two losses and the gradient of every parameter.
"""
from __future__ import annotations

import sys
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.golden import _ref_shim  # noqa: E402
from tests.golden.make_golden import seed  # noqa: E402
from tests.golden.make_golden_coca import POOL96, randomize, SMALL  # noqa: E402

OUT = Path(__file__).resolve().parent
warnings.filterwarnings("ignore")


def main():
    _ref_shim.install()
    from torchmultimodal.models.coca.coca_model import coca_vit, CoCaForPretraining

    st = {}
    for prefix, kw, seed_v, fixture in (("s64.", SMALL, 51, "coca_small.npz"), ("p96.", POOL96, 53, "coca_pool96.npz")):
        z = np.load(OUT / fixture)
        seed(seed_v)
        model = coca_vit(**kw, cascaded_pooler=False)
        randomize(model, torch.Generator().manual_seed(seed_v + 1))
        pre = CoCaForPretraining(model).train()
        losses = pre(torch.from_numpy(z["par.images"]), torch.from_numpy(z["par.texts"]))
        (losses["contrastive"] + losses["captioning"]).backward()
        st[prefix + "contrastive"], st[prefix + "captioning"] = losses["contrastive"].detach().numpy(), losses["captioning"].detach().numpy()
        none = []
        for k, p in pre.named_parameters():
            if p.grad is None:
                none.append(k)
            elif prefix == "s64." or "pooler" in k or "vision_proj" in k or k.endswith("ln_final.weight"):
                # float16 storage (the parity tolerance is percent-level); the 96-wide model keeps only what differs: the pooler path
                st[prefix + "g." + k] = p.grad.numpy().astype(np.float16) if p.grad.abs().max() < 6e4 else p.grad.numpy()
        st[prefix + "no_grad_keys"] = np.array(none)
        print(prefix, {k: float(v) for k, v in losses.items()}, "no grad:", none)
    np.savez_compressed(OUT / "coca_grad.npz", **st)
    print("written", len(st), "arrays")


if __name__ == "__main__":
    main()
