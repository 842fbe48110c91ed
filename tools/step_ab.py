"""Same-box, same-process A/B of the headline step (CLIP ViT-B/16, B = 256, forward + contrastive loss) under different settings,
alternating arms (guide rules 13 / 24).  Each arm is timed three ways: eager with HIP events (what bench.py reports), the host time to
ENQUEUE a step (is the Python thread the limit?), and a HIP-graph replay of the same step (no host in the loop).

    python tools/step_ab.py --arms ring,r02attn [--rounds 3] [--steps 20]
arms (joined with +): base, r02attn (register-staged attention kernel), streams,
ln_cached (LayerNorm input loads never non-temporal), stagger<percent> (the delta_ln and phases2 arms of r03 / r04 were retired in r05),
lead<n> (launches of half 0 before half 1 starts), slack<percent> (slack-aware start-up stagger of the grouped GEMM), gm<4|8> (its tile-order group), cn<k> (column tiles per chunk of the tile order; cn-1 = none)"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arms", default="ring,r02attn")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--bf16-images", action="store_true", help="feed an already-cast bf16 batch (default: fp32 images like bench.py; the cast is part of the step)")
    a = ap.parse_args()
    from multimodal_amd.models.clip import clip_vit_b16
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.synthetic import clip_batch

    L = _lib.lib()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = clip_vit_b16().to(dev).eval()
    loss_fn = ContrastiveLossWithTemperature().to(dev)
    images, ids = clip_batch(a.batch)
    images, ids = images.to(dev), ids.to(dev)
    if a.bf16_images:
        images = images.to(torch.bfloat16)

    from multimodal_amd.schedule import set_schedule

    def set_arm(name):
        L.mmamd_debug_set_attn_variant(0)
        L.mmamd_debug_set_attn_variant(3100)
        L.mmamd_debug_set_attn_variant(3110)
        L.mmamd_debug_set_gemm_stagger(60)
        L.mmamd_debug_set_gemm_knob(0, 0)
        L.mmamd_debug_set_gemm_knob(1, 0)
        L.mmamd_debug_set_gemm_knob(4, 0)
        set_schedule(two_tower="auto")
        for part in name.split("+"):
            if part in ("ring", "base"):
                pass
            elif part == "r02attn":
                L.mmamd_debug_set_attn_variant(1000)
            elif part == "streams":
                set_schedule(two_tower="streams")
            elif part.startswith("stagger"):  # start-up stagger of the persistent GEMMs, per cent of a tile time (default 60)
                L.mmamd_debug_set_gemm_stagger(int(part[7:]))
            elif part.startswith("slack"):
                L.mmamd_debug_set_gemm_knob(1, int(part[5:]))
            elif part.startswith("gm"):
                L.mmamd_debug_set_gemm_knob(0, int(part[2:]))
            elif part.startswith("cn"):  # column tiles per chunk of the persistent GEMMs' tile order (r06; cn-1 = no chunking, default = by W size)
                L.mmamd_debug_set_gemm_knob(4, int(part[2:]))
            elif part == "ln_rev":  # grouped LayerNorm walks its rows from the last to the first (the rows the GEMM before it wrote last are read first)
                L.mmamd_debug_set_attn_variant(3111)
            elif part == "ln_cached":  # LayerNorm input loads never non-temporal (the r02 behaviour)
                L.mmamd_debug_set_attn_variant(3101)
            else:
                raise SystemExit(f"unknown arm {part}")

    def step():
        out = model(images, ids)
        return loss_fn(out.embeddings_a, out.embeddings_b)

    res = {}
    arms = a.arms.split(",")
    with torch.no_grad():
        ref = None
        for arm in arms:  # results of every arm against the first one
            set_arm(arm)
            out = model(images, ids)
            ea, eb = out.embeddings_a.float().clone(), out.embeddings_b.float().clone()
            if ref is None:
                ref = (ea, eb)
            print(f"{arm}: loss {float(loss_fn(out.embeddings_a, out.embeddings_b)):.6f}  max|d emb_a| {float((ea - ref[0]).abs().max()):.3e}  "
                  f"max|d emb_b| {float((eb - ref[1]).abs().max()):.3e} vs {arms[0]}", flush=True)
        for rnd in range(a.rounds):
            for arm in arms:
                set_arm(arm)
                for _ in range(3):
                    loss = step()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record()
                for _ in range(a.steps):
                    loss = step()
                e1.record()
                t_enq = time.perf_counter() - t0
                torch.cuda.synchronize()
                r = {"eager_ms": e0.elapsed_time(e1) / a.steps, "host_enqueue_ms": t_enq / a.steps * 1e3, "loss": float(loss)}
                if not a.no_graph:
                    s = torch.cuda.Stream()
                    with torch.cuda.stream(s):
                        step()
                        torch.cuda.synchronize()
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=s):
                            gl = step()
                        for _ in range(3):
                            g.replay()
                        torch.cuda.synchronize()
                        e0.record()
                        for _ in range(a.steps):
                            g.replay()
                        e1.record()
                        torch.cuda.synchronize()
                        r["graph_ms"] = e0.elapsed_time(e1) / a.steps
                        r["graph_loss"] = float(gl)
                    del g
                res.setdefault(arm, []).append(r)
                print(arm, json.dumps({k: round(v, 4) for k, v in r.items()}), flush=True)
    set_arm("base")
    for arm in arms:
        for key in ("eager_ms", "host_enqueue_ms", "graph_ms"):
            v = sorted(x[key] for x in res[arm] if key in x)
            if v:
                print(f"{arm:12s} {key:16s} median {v[len(v) // 2]:7.3f}  min {v[0]:7.3f}  max {v[-1]:7.3f}")


if __name__ == "__main__":
    main()
