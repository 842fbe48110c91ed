#!/usr/bin/env python
"""Max errors of the HIP path vs the reference fixtures (full-size CLIP, seed-0 init, synthetic batch)."""
import math, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import multimodal_amd.models.clip as mc
from multimodal_amd.modules.losses.contrastive_loss_with_temperature import contrastive_loss_with_temperature
from multimodal_amd.utils.synthetic import clip_batch
for name, factory, B in (("clip_b32_b8", "clip_vit_b32", 8), ("clip_b16_b4", "clip_vit_b16", 4)):
    z = np.load(ROOT / "tests" / "golden" / f"{name}.npz")
    torch.manual_seed(0)
    model = getattr(mc, factory)().cuda().eval()
    images, ids = clip_batch(B)
    with torch.no_grad():
        out = model(images.cuda(), ids.cuda())
        scale = torch.nn.Parameter(torch.tensor(math.log(1 / 0.07), device="cuda"))
        lo = contrastive_loss_with_temperature(out.embeddings_a, out.embeddings_b, scale)
    ea = np.abs(out.embeddings_a.cpu().numpy() - z["emb_a"]).max(); eb = np.abs(out.embeddings_b.cpu().numpy() - z["emb_b"]).max()
    la = np.abs(lo.logits_a.cpu().numpy() - z["logits_a"]).max(); dl = abs(float(lo.loss) - float(z["loss"]))
    print(f"{name}: max|d emb_a| {ea:.2e}  max|d emb_b| {eb:.2e}  max|d logits| {la:.3e}  |d loss| {dl:.2e}")
