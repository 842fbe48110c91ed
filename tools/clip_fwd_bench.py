"""Forward + contrastive loss timing of any CLIP factory on one MI355X (bench.py is fixed to the headline ViT-B/16 config):
    python tools/clip_fwd_bench.py --model l14 [--batch 256] [--steps 10]
ViT-L/14 at B = 256 is the per-GPU shape of SURVEY.md section 8 cfg 3 (175.33 GF/pair)."""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
GF = {"b16": 41.09, "b32": 14.78, "l14": 175.33}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="l14", choices=list(GF))
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--vision-only", action="store_true", help="time the image tower alone (north-star target: >= 40 %% MFMA on ViT-B/16 attention+MLP)")
    a = ap.parse_args()
    from multimodal_amd.models.clip import model as M
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.synthetic import clip_batch

    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    model = getattr(M, {"b16": "clip_vit_b16", "b32": "clip_vit_b32", "l14": "clip_vit_l14"}[a.model])().to(dev).eval()
    loss_fn = ContrastiveLossWithTemperature().to(dev)
    images, ids = clip_batch(a.batch)
    images, ids = images.to(dev), ids.to(dev)

    VISION_GF = {"b16": 35.127, "b32": 8.82, "l14": 162.03}

    def step():
        with torch.no_grad():
            if a.vision_only:
                return model.encoder_a(images).sum()
            out = model(images, ids)
            return loss_fn(out.embeddings_a, out.embeddings_b)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(a.steps):
        loss = step()
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / a.steps
    gf = VISION_GF[a.model] if a.vision_only else GF[a.model]
    print(json.dumps({"workload": f"CLIP ViT-{a.model.upper()} " + ("image tower forward" if a.vision_only else "fwd + contrastive loss"),
                      "batch": a.batch, "ms_per_step": round(ms, 3),
                      "pairs_per_s": round(a.batch / ms * 1e3, 1), "gflop_per_pair": gf,
                      "tflops": round(a.batch * gf / ms, 1), "mfma_frac": round(a.batch * gf / ms / 2500.0, 4),
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "loss": float(loss)}))


if __name__ == "__main__":
    main()
