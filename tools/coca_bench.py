"""CoCa ViT-L/14 forward + pre-training losses on one MI355X — SURVEY.md section 8 cfg 5 per-GPU shape (B = 128, parallel pooler:
coca_vit(**l14 kwargs, cascaded_pooler=False); the reference's CoCaForPretraining fails on the cascaded pooler).
    python tools/coca_bench.py [--batch 128] [--steps 5]
Algorithmic FLOPs: ~205 GF/sample.  Prints one JSON line."""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

L14 = dict(vision_patch_size=14, vision_n_layer=24, vision_n_head=16, vision_dim_feedforward=4096, vision_include_cls_embed=False,
           vocab_size=49408, num_text_positions=77, text_hidden_dim=768, text_n_layer=12, text_n_head=12, text_dim_feedforward=3072,
           text_output_dim=768, fusion_n_layer=12, fusion_n_head=12, fusion_dim_feedforward=3072, multimodal_output_projection_dim=49408,
           pooler_input_embed_dim=1024, pooler_output_embed_dim=768, pooler_n_head=8, cascaded_pooler=False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--train", action="store_true", help="training step: forward + both losses + backward + SGD")
    a = ap.parse_args()
    from multimodal_amd.models.coca.coca_model import coca_vit, CoCaForPretraining

    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    model = CoCaForPretraining(coca_vit(**L14)).to(dev)
    model = model.train() if a.train else model.eval()
    opt = torch.optim.SGD(model.parameters(), lr=1e-4) if a.train else None
    B = a.batch
    g = torch.Generator().manual_seed(1)
    images = torch.randn(B, 3, 224, 224, generator=g).to(dev)
    texts = torch.randint(1, 49408, (B, 77), generator=g)
    texts[:, 60:] = 0
    texts = texts.to(dev)

    def step():
        if a.train:
            opt.zero_grad(set_to_none=True)
            losses = model(images, texts)
            (losses["contrastive"] + losses["captioning"]).backward()
            opt.step()
            return {k: v.detach() for k, v in losses.items()}
        with torch.no_grad():
            return model(images, texts)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(a.steps):
        r = step()
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / a.steps
    gf = 205.0 * (3 if a.train else 1)
    print(json.dumps({"workload": ("TRAINING step: " if a.train else "") + "CoCaForPretraining(coca_vit L/14, parallel pooler) fwd + losses" + (" + bwd + SGD" if a.train else ""), "batch": B,
                      "ms_per_step": round(ms, 3), "samples_per_s": round(B / ms * 1e3, 1), "gflop_per_sample": gf,
                      "tflops": round(B * gf / ms, 1), "mfma_frac": round(B * gf / ms / 2500.0, 4),
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                      "contrastive": float(r["contrastive"]), "captioning": float(r["captioning"])}))


if __name__ == "__main__":
    main()
