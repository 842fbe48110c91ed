#!/usr/bin/env python
"""bench.py — image-text pairs/sec of the dual-encoder contrastive forward + loss on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): CLIP ViT-B/16 + text transformer,
per-GPU batch 256, 224x224 synthetic images + 77 synthetic token ids, forward of both towers + L2 normalise +
ContrastiveLossWithTemperature (local loss at N=1; global loss with one packed RCCL all-gather at N>1).
One "step" = one such pass over one batch that is already resident in HBM.  Weak scaling: every rank encodes its
own 256 pairs; `value` = N*256*K / max-over-ranks wall time of the K timed steps.

Extra objects on the JSON line:
  roofline      the dominant kernel = the MLP-up GEMM of the vision tower ([50432 x 3072 x 768], 238 GFLOP per launch,
                12 launches per step): algorithmic FLOPs / its mean launch duration, measured with HIP events on the
                launch stream inside the timed region, against the 2.5 PFLOP/s dense bf16 MFMA peak.
  step_mfma_frac  whole-step figure: pairs/s x 41.09 GFLOP/pair / 2.5 PFLOP/s.
  cpu_baseline  the numpy oracle (a port, not the reference) on the host cores, on a bounded sample (B=32) of the
                same workload; rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 (MI355X_MICROARCH.md "Chip-level parameters")
GF_PER_PAIR = 41.09             # BASELINE.md §3, CLIP ViT-B/16 + text, full S^2 attention counted


def _blas_threads() -> int:
    """Threads the numpy BLAS actually uses (OpenBLAS caps at its build-time MAX_THREADS, not os.cpu_count())."""
    try:
        from threadpoolctl import threadpool_info

        n = [p.get("num_threads", 0) for p in threadpool_info() if p.get("user_api") == "blas"]
        if n:
            return int(max(n))
    except Exception:
        pass
    return os.cpu_count() or 1


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (the metric is quoted at 256)")
    ap.add_argument("--cpu-sample", type=int, default=32, help="batch of the CPU-baseline sample; 0 = skip")
    ap.add_argument("--gemm-variant", type=int, default=0)
    ap.add_argument("--no-probe", action="store_true", help="do not bracket the dominant GEMM with events")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU path to measure")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # "nccl" IS RCCL on ROCm; intra-node transport = xGMI

    from multimodal_amd import build

    if local_rank == 0:
        build.build()
    if world > 1:
        dist.barrier()

    from multimodal_amd import ops
    from multimodal_amd.models.clip import clip_vit_b16
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.synthetic import clip_batch

    ops.set_gemm_variant(args.gemm_variant)
    torch.manual_seed(0)
    model = clip_vit_b16()
    sd_host = {k: v.numpy() for k, v in model.state_dict().items()} if (rank == 0 and world == 1 and args.cpu_sample) else None
    model = model.to(dev).eval()
    loss_fn = ContrastiveLossWithTemperature().to(dev)
    B = args.batch
    images, ids = clip_batch(B, rank=rank)
    images_d, ids_d = images.to(dev), ids.to(dev)

    def step():
        out = model(images_d, ids_d)
        return loss_fn(out.embeddings_a, out.embeddings_b)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    S_img = 197
    probe = ops.GemmProbe(B * S_img, 3072, 768)
    with torch.no_grad():
        for _ in range(args.warmup):
            loss = step()
        fence()
        t0 = time.perf_counter()
        if args.no_probe:
            for _ in range(args.steps):
                loss = step()
        else:
            with probe:
                for _ in range(args.steps):
                    loss = step()
        fence()
        dt = time.perf_counter() - t0
    loss_val = float(loss)
    if not math.isfinite(loss_val):
        raise SystemExit(f"non-finite loss {loss_val}")

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t)
    pairs_per_s = world * B * args.steps / dt_max

    # the same kernel alone on the GPU (the timed region overlaps the text tower on a side stream, which stretches the
    # in-region duration of every vision kernel): reported next to the in-region figure as "isolated"
    iso_ms = None
    if not args.no_probe:
        M, N, K = probe.shape
        a_ = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w_ = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        b_ = torch.randn(N, device=dev)
        o_ = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            ops.gemm_bf16(a_, w_, b_, act=ops.ACT_QUICKGELU, out=o_)
        torch.cuda.synchronize(dev)
        tm = ops.StreamTimer()
        tm.start()
        for _ in range(10):
            ops.gemm_bf16(a_, w_, b_, act=ops.ACT_QUICKGELU, out=o_)
        tm.stop()
        iso_ms = tm.elapsed_ms() / 10
        del a_, w_, b_, o_

    roofline = None
    durs = [] if args.no_probe else probe.durations_ms()
    if durs:
        mean_ms = sum(durs) / len(durs)
        flops = 2.0 * (B * S_img) * 3072 * 768
        achieved = flops / (mean_ms * 1e-3) / 1e12
        traffic = None
        pmc = ROOT / "profiles" / "pmc_dominant_kernel.json"
        if pmc.exists():
            try:
                traffic = json.loads(pmc.read_text()).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "mfma", "kernel": f"gemm_bf16_nt MLP-up [{B * S_img}x3072x768] (+bias, QuickGELU)",
                    "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic,
                    "launch_ms": round(mean_ms, 4), "launches_timed": len(durs),
                    "algorithmic_flops_per_launch": flops,
                    "isolated": {"launch_ms": round(iso_ms, 4), "achieved": round(flops / (iso_ms * 1e-3) / 1e12, 2),
                                 "frac": round(flops / (iso_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}}

    cpu_baseline = None
    if sd_host is not None:
        from oracle import clip_oracle as oc

        n = args.cpu_sample
        im_s, id_s = images[:n].numpy(), ids[:n].numpy()
        tc = time.perf_counter()
        a, b = oc.clip_forward(sd_host, im_s, id_s, 12, 8)
        o = oc.contrastive_loss_with_temperature(a, b, math.log(1 / 0.07))
        tcpu = time.perf_counter() - tc
        cpu_baseline = {"value": round(n / tcpu, 3), "unit": "pairs/s", "cores": _blas_threads(), "kind": "port",
                        "sample": f"first {n} pairs of the same synthetic ViT-B/16 batch, fp32 numpy+OpenBLAS oracle, "
                                  f"1 pass, {tcpu:.1f} s", "loss_on_sample": round(float(o["loss"]), 5)}

    if rank == 0:
        line = {
            "metric": "image-text pairs/sec (fwd+contrastive loss), CLIP ViT-B/16 B=256",
            "value": round(pairs_per_s, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "CLIP ViT-B/16 + text transformer forward + ContrastiveLossWithTemperature "
                                   f"({'local' if world == 1 else 'global, packed RCCL all-gather'}), random-init weights",
                       "per_gpu_batch": B, "global_batch": world * B, "seq_img": S_img, "seq_txt": 77,
                       "parallelism": f"dp{world}", "gemm_variant": args.gemm_variant},
            "loss": round(loss_val, 5),
            "step_mfma_frac": round(pairs_per_s / world * GF_PER_PAIR * 1e9 / (MFMA_BF16_PEAK_TFLOPS * 1e12), 4),
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
