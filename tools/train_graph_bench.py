"""CLIP ViT-B/16 training step (B = 256): eager vs torch.compile(fullgraph=True) vs torch.compile(mode="reduce-overhead") (HIP-graph replay of the
compiled forward and backward graphs; .backward() and the optimizer step are called outside).   python tools/train_graph_bench.py"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    from multimodal_amd.models.clip import clip_vit_b16
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.synthetic import clip_batch

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = clip_vit_b16().to(dev).train()
    loss_fn = ContrastiveLossWithTemperature().to(dev)
    params = list(model.parameters()) + list(loss_fn.parameters())
    opt = torch.optim.SGD(params, lr=1e-4)
    images, ids = clip_batch(256)
    images, ids = images.to(dev), ids.to(dev)

    def fwd(images, ids):
        out = model(images, ids)
        return loss_fn(out.embeddings_a, out.embeddings_b)

    def run(fn, n=5, warm=3):
        for _ in range(warm):
            opt.zero_grad(set_to_none=True)
            fn(images, ids).backward()
            opt.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            opt.zero_grad(set_to_none=True)
            loss = fn(images, ids)
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, float(loss)

    print("eager            %.2f ms  loss %.4f" % run(fwd), flush=True)
    c1 = torch.compile(fwd, fullgraph=True)
    print("compile          %.2f ms  loss %.4f" % run(c1), flush=True)
    try:
        c2 = torch.compile(fwd, fullgraph=True, mode="reduce-overhead")
        print("reduce-overhead  %.2f ms  loss %.4f" % run(c2, warm=5), flush=True)
    except Exception as e:  # noqa: BLE001
        print("reduce-overhead failed:", type(e).__name__, str(e)[:300])
    print("eager again      %.2f ms  loss %.4f" % run(fwd), flush=True)


if __name__ == "__main__":
    main()
