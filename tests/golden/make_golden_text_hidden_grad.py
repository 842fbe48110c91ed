"""Gradients through CLIPTextEncoder(..., return_hidden_state=True) from the REFERENCE (models/clip/text_encoder.py:113-127):
python -m tests.golden.make_golden_text_hidden_grad  ->  text_hidden_grad.npz
  a small text tower (width 128, 2 layers, 2 heads, vocabulary 1000, context 77) in TRAIN mode: hidden = ln_final(encoder(...)) [4, 77, 128],
  loss = sum(hidden * w) -> every parameter gradient (the projection gets none on this path)."""
from __future__ import annotations

import sys
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.golden import _ref_shim  # noqa: E402
from tests.golden.make_golden import sd_np, seed  # noqa: E402

OUT = Path(__file__).resolve().parent
warnings.filterwarnings("ignore")


def tnp(t):
    return t.detach().numpy().copy()


def main():
    _ref_shim.install()
    from torchmultimodal.models.clip.text_encoder import CLIPTextEncoder

    torch.set_num_threads(8)
    seed(81)
    enc = CLIPTextEncoder(embedding_dim=64, context_length=77, vocab_size=1000, width=128, dim_feedforward=256, heads=2, layers=2).train()
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(1, 999, (4, 77), generator=g)
    w = torch.randn(4, 77, 128, generator=g)
    hidden = enc(ids, return_hidden_state=True)
    (hidden * w).sum().backward()
    st = {"ids": tnp(ids), "w": tnp(w), "hidden": tnp(hidden)}
    st.update({"sd." + k: v for k, v in sd_np(enc).items()})
    st.update({"g." + k: tnp(p.grad) for k, p in enc.named_parameters() if p.grad is not None})
    np.savez_compressed(OUT / "text_hidden_grad.npz", **st)
    print("text_hidden_grad.npz", hidden.shape, len([k for k in st if k.startswith("g.")]), "gradients; no gradient:",
          [k for k, p in enc.named_parameters() if p.grad is None])


if __name__ == "__main__":
    main()
