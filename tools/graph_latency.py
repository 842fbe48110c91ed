"""Small-batch serving latency of CLIP ViT-B/16 forward (both towers + normalise), eager launches vs one captured HIP graph:
    python tools/graph_latency.py [--batches 1,8,32]
The C-ABI neither allocates nor synchronises, so the whole forward is capturable with torch.cuda.CUDAGraph (hipGraph underneath)."""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def timeit(fn, iters):
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,8,32")
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    from multimodal_amd.models.clip import clip_vit_b16
    from multimodal_amd.utils.synthetic import clip_batch

    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    model = clip_vit_b16().to(dev).eval()
    rows = []
    for B in (int(x) for x in a.batches.split(",")):
        images, ids = clip_batch(B)
        images, ids = images.to(dev), ids.to(dev)
        with torch.no_grad():
            for _ in range(3):
                ref = model(images, ids)
            eager = timeit(lambda: model(images, ids), a.iters)
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                model(images, ids)
                with torch.cuda.graph(g, stream=s):
                    out = model(images, ids)
            torch.cuda.current_stream().wait_stream(s)
            g.replay()
            torch.cuda.synchronize()
            same = bool(torch.equal(out.embeddings_a, ref.embeddings_a) and torch.equal(out.embeddings_b, ref.embeddings_b))
            graph = timeit(g.replay, a.iters)
        rows.append({"batch": B, "eager_ms": round(eager, 3), "graph_ms": round(graph, 3), "speedup": round(eager / graph, 2), "bit_identical": same})
    print(json.dumps({"workload": "CLIP ViT-B/16 forward, eager vs captured graph", "rows": rows}))


if __name__ == "__main__":
    main()
