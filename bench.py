#!/usr/bin/env python
"""bench.py — image-text pairs/sec of the dual-encoder contrastive forward + loss on MI355X.

    python bench.py --gpus N --steps K --warmup W
        N = 1: runs in this process.  N > 1 without a launcher environment (WORLD_SIZE unset): re-executes itself under
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` —
        one process per GPU over RCCL; rank 0 prints the ONE JSON line.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
        the same, launched by the caller (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).
    python bench.py --gpus 2 --backend gloo --dry-run
        control flow of the N > 1 path on CPU (launcher, rendezvous, barrier + max-over-ranks timing, the packed all-gather
        and its layout check, the JSON line) with stand-in features instead of the towers: what tests/test_bench_launcher.py
        runs.  Not a measurement: the line says "dry_run": true and carries no roofline.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): CLIP ViT-B/16 + text transformer,
per-GPU batch 256, 224x224 synthetic images + 77 synthetic token ids, forward of both towers + L2 normalise +
ContrastiveLossWithTemperature (local loss at N=1; global loss with one packed RCCL all-gather at N>1).
One "step" = one such pass over one batch that is already resident in HBM.  Weak scaling: every rank encodes its
own 256 pairs; `value` = N*256*K / max-over-ranks wall time of the K timed steps (warm-up 10, 50 timed steps by
default; `ms_per_step_median` = the median of the per-step HIP-event durations of rank 0, reported next to the mean).

Inputs are what the reference's transform hands over: fp32 images and int64 ids, resident in HBM; the fp32 -> bf16 cast of the image batch is
part of the timed step (`--bf16-images` feeds an already-cast batch instead; the line carries that figure too, as `bf16_resident_images`).

Extra objects on the JSON line:
  roofline      the dominant kernel = the one with the LARGEST SHARE of the step: gemm_bf16_nt_kernel_ppg<true,0>, the grouped fp32-residual
                GEMM of both towers (out-projection [B*197 x 768 x 768] and MLP-down [B*197 x 768 x 3072] + the text tower's, 24 launches per
                step, ~37 % of it): algorithmic FLOPs of the timed launches / their summed duration, measured with HIP events on the launch
                stream(s) inside the timed region, against the 2.5 PFLOP/s dense bf16 MFMA peak; `by_shape` splits the two shapes;
                `other_kernels` carries the MLP-up (+QuickGELU) and qkv grouped GEMMs, timed the same way in a few extra steps AFTER the region.
  step_mfma_frac  whole-step figure: pairs/s x 41.09 GFLOP/pair / 2.5 PFLOP/s.
  cpu_baseline  rank 0, N=1 only: the reference's CPU path on this box's host cores, on a bounded sample (B=32, 1 warm-up +
                3 timed passes) of the same batch.  /root/reference does not exist on the GPU box, so what is timed here is
                oracle/torch_cpu_clip.py — the reference's module composition restated on the same torch.nn modules, i.e. the
                same ATen CPU kernels ("kind": "port" in the contract's vocabulary, "kind_detail" says what it is); the figure of the reference ITSELF, measured in the
                build container by tests/golden/make_golden_headline.py, is carried next to it ("reference_itself").
  rccl_ranks_seen  N > 1: all_reduce(SUM) of a one per rank over the bench's process group (RCCL for --backend nccl).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 (MI355X_MICROARCH.md "Chip-level parameters")
GF_PER_PAIR = 41.09             # BASELINE.md §3, CLIP ViT-B/16 + text, full S^2 attention counted
METRIC = "image-text pairs/sec (fwd+contrastive loss), CLIP ViT-B/16 B=256"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (the metric is quoted at 256)")
    ap.add_argument("--cpu-sample", type=int, default=32, help="batch of the CPU-baseline sample; 0 = skip")
    ap.add_argument("--cpu-full", type=int, default=1, help="1: also ONE timed CPU pass over the whole batch (about 17 s at B = 256) next to the sample's figure")
    ap.add_argument("--gemm-variant", type=int, default=0)
    ap.add_argument("--no-probe", action="store_true", help="do not bracket the dominant GEMM with events")
    ap.add_argument("--fp32-images", action="store_true", help="(default since r04) feed fp32 images: the stem converts them to bf16 inside the timed step")
    ap.add_argument("--bf16-images", action="store_true", help="feed an already-cast bf16 image batch (the r03 default): the cast is then outside the timed step")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help='"nccl" IS RCCL on ROCm; gloo only with --dry-run')
    ap.add_argument("--dry-run", action="store_true", help="CPU control-flow run of the multi-rank path (no towers, no measurement)")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step (both towers, loss, and for N > 1 the packed all-gather) in ONE HIP graph and time K replays: no Python "
                         "between the kernels, so host jitter cannot skew N lock-stepped ranks; falls back to eager launches (and says so) when "
                         "the capture fails")
    ap.add_argument("--no-affinity", action="store_true", help="do not pin each rank to the CPUs next to its GPU")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST MODE, not a measurement: every rank of an N > 1 job runs on device 0 and the process group is gloo (RCCL refuses two "
                         "ranks on one device) -- the driver's N = 8 control flow with the real towers on a one-GPU box (tests/test_gpu_loss_w8.py)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group even with ONE rank: the N > 1 code path (communicator, packed all-gather, rank-offset "
                         "labels, fences, max-over-ranks) on a single-GPU box")
    return ap.parse_args(argv)


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def relaunch(args) -> int:
    """--gpus N > 1 outside a launcher: become `torch.distributed.run` with N local ranks (same argv)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_rank_to_local_cpus(local_rank: int, world: int, bdf=None):
    """One process per GPU: keep each rank's Python thread (and its helper threads) on its own slice of the CPUs of the NUMA node its GPU
    hangs off (/sys/bus/pci/devices/<bdf>/local_cpulist), so that 8 launch loops do not migrate across sockets or share cores.  Returns a
    description for the JSON line; never fatal."""
    if not hasattr(os, "sched_setaffinity"):
        return None
    try:
        avail = sorted(os.sched_getaffinity(0))
        local = avail
        where = "all"
        if bdf:
            path = f"/sys/bus/pci/devices/{bdf}/local_cpulist"
            if os.path.exists(path):
                near = [c for c in _parse_cpulist(open(path).read()) if c in set(avail)]
                if near:
                    local, where = near, f"numa-local to {bdf}"
        per = len(local) // max(world, 1)
        if per < 4:  # too few cores for a private slice (the launch thread, the HIP runtime's and RCCL's helpers need several): share the list
            mine, where = local, where + ", shared"
        else:
            lo = (local_rank * per) % len(local)
            mine = local[lo:lo + per] or local
        os.sched_setaffinity(0, mine)
        return {"cpus": f"{mine[0]}-{mine[-1]}" if mine == list(range(mine[0], mine[-1] + 1)) else ",".join(map(str, mine)),
                "count": len(mine), "policy": where}
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def fence(dev, dist, world):
    import torch

    if dev is not None:
        torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
        if dev is not None:
            torch.cuda.synchronize(dev)


def dry_main(args, rank, world) -> None:
    """The N > 1 control flow on CPU tensors: rendezvous, ranks-seen reduction, packed gather + layout check, fenced timing, JSON."""
    import torch
    import torch.distributed as dist

    from multimodal_amd.utils.distributed import gather_packed_features

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend)
    ones = torch.ones(1)
    if world > 1:
        dist.all_reduce(ones)
    affinity = None if args.no_affinity else pin_rank_to_local_cpus(int(os.environ.get("LOCAL_RANK", "0")), world)
    B, E = args.batch, 512
    g = torch.Generator().manual_seed(1234 + rank)
    a = torch.nn.functional.normalize(torch.randn(B, E, generator=g))
    b = torch.nn.functional.normalize(torch.randn(B, E, generator=g))
    for _ in range(args.warmup):
        gather_packed_features(a, b)
    fence(None, dist, world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        buf, r, w = gather_packed_features(a, b)
    fence(None, dist, world)
    dt = time.perf_counter() - t0
    assert (r, w) == (rank, world) and buf.shape == (world * B, 2 * E)
    assert torch.equal(buf[rank * B:(rank + 1) * B, :E], a) and torch.equal(buf[rank * B:(rank + 1) * B, E:], b)
    t = torch.tensor([dt], dtype=torch.float64)
    per_rank = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, t.clone())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    else:
        per_rank = [t.clone()]
    if rank == 0:
        print(json.dumps({"metric": METRIC, "dry_run": True, "value": None, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(float(t) / max(args.steps, 1) * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic", "backend": args.backend,
                          "ranks_seen": int(ones.item()), "cpu_affinity_rank0": affinity,
                          "per_rank_ms_per_step": [round(float(x) / max(args.steps, 1) * 1e3, 3) for x in per_rank],
                          "config": {"workload": "control flow of the multi-rank path on CPU tensors (no towers)", "per_gpu_batch": B,
                                     "global_batch": world * B, "parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline_leg(sd_host, images, ids, n, full=False):
    """The reference's CPU path, timed on this host (see the module docstring): 1 warm-up + 3 timed passes of n pairs."""
    import torch

    from oracle.torch_cpu_clip import TorchCPUCLIP

    m = TorchCPUCLIP(sd_host, vision_heads=12, text_heads=8)
    im_s, id_s = images[:n].clone(), ids[:n].clone()
    # thread count: torch's default on a 256-logical-CPU host (128 threads) is slower at this batch than a few dozen threads
    # (measured: 4.7 pairs/s at 128); one pass per candidate doubles as the warm-up, the fastest is used for the timed passes
    ncpu = os.cpu_count() or 1
    sweep = {}
    for t in sorted({c for c in (8, 16, 32, 64) if c <= ncpu} or {ncpu}):
        torch.set_num_threads(t)
        tc = time.perf_counter()
        m.forward_loss(im_s, id_s)
        sweep[t] = time.perf_counter() - tc
    best_t = min(sweep, key=sweep.get)
    torch.set_num_threads(best_t)
    times = []
    for _ in range(3):
        tc = time.perf_counter()
        out = m.forward_loss(im_s, id_s)
        times.append(time.perf_counter() - tc)
    med = sorted(times)[1]
    full_batch = None
    if full and images.shape[0] > n:  # ONE pass over the whole batch at the same thread count (carry both figures)
        tc = time.perf_counter()
        out_full = m.forward_loss(images, ids)
        tf = time.perf_counter() - tc
        full_batch = {"value": round(images.shape[0] / tf, 3), "unit": "pairs/s", "pairs": int(images.shape[0]), "seconds": round(tf, 2), "passes": 1,
                      "threads": best_t, "loss": round(float(out_full[4]), 5)}
    ref_itself = None
    p = ROOT / "profiles" / "r02_reference_cpu.json"
    if p.exists():
        try:
            z = json.loads(p.read_text())["clip_b16_b256"]
            ref_itself = {"value": z["pairs_per_s"], "unit": "pairs/s", "cores": z["cores"], "threads": z["threads"], "batch": z["batch"],
                          "where": "build container, tests/golden/make_golden_headline.py (the imported reference, fp32)"}
        except Exception:
            ref_itself = None
    return {"value": round(n / med, 3), "unit": "pairs/s", "cores": torch.get_num_threads(), "host_logical_cpus": os.cpu_count(),
            "kind": "port", "kind_detail": "oracle/torch_cpu_clip.py: the reference's module composition restated on the same torch.nn CPU modules",
            "thread_sweep_s": {str(k): round(v, 2) for k, v in sweep.items()},
            "sample": f"first {n} pairs of the same synthetic ViT-B/16 batch, fp32, torch {torch.__version__} CPU kernels through "
                      f"nn.TransformerEncoder (oracle/torch_cpu_clip.py), one warm-up pass per thread count then 3 timed passes at "
                      f"{best_t} threads, median {med:.2f} s",
            "loss_on_sample": round(float(out[4]), 5), "full_batch": full_batch, "reference_itself": ref_itself}


def main() -> None:
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(relaunch(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_main(args, rank, world)
    if args.backend != "nccl":
        raise SystemExit("the measured path runs over RCCL (--backend nccl); gloo is for --dry-run")
    if args.share_gpu:
        local_rank = 0  # every rank on the one device of the box; gloo carries the collectives
        args.no_probe = True
        args.cpu_sample = 0

    # stdout carries exactly ONE line, the JSON record: everything else that writes to fd 1 from here on (RCCL prints its library path
    # through C stdio, flushed at exit, i.e. AFTER the record) goes to stderr; the record is written to the saved descriptor
    sys.stdout.flush()
    record_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU path to measure")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no device ({torch.cuda.device_count()} visible); one process per GPU")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    affinity = None
    if not args.no_affinity and world > 1:
        bdf = None
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:  # noqa: BLE001
            bdf = None
        affinity = pin_rank_to_local_cpus(int(os.environ.get("LOCAL_RANK", "0")), world, bdf)

    import torch.distributed as dist

    ranks_seen = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # --force-dist without a launcher
            os.environ["MASTER_PORT"] = str(free_port())
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # "nccl" IS RCCL on ROCm; intra-node transport = xGMI
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())

    from multimodal_amd import build

    if rank == 0 or (local_rank == 0 and not args.share_gpu):
        build.build()
    if use_dist:
        dist.barrier()

    from multimodal_amd import ops
    from multimodal_amd.models.clip import clip_vit_b16
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.synthetic import clip_batch

    ops.set_gemm_variant(args.gemm_variant)
    torch.manual_seed(0)
    model = clip_vit_b16()
    sd_host = {k: v.clone() for k, v in model.state_dict().items()} if (rank == 0 and world == 1 and args.cpu_sample) else None
    model = model.to(dev).eval()
    loss_fn = ContrastiveLossWithTemperature().to(dev)
    B = args.batch
    images, ids = clip_batch(B, rank=rank)
    # resident inputs as the reference's transform hands them over: fp32 images, int64 ids (the cast to bf16 belongs to the step;
    # --bf16-images restores the r03 form, where the batch was cast once outside the timed region -- same rounding, same results)
    images_d, ids_d = images.to(dev), ids.to(dev)
    images_bf16 = images_d.to(torch.bfloat16)
    if args.bf16_images:
        images_d = images_bf16

    def step():
        out = model(images_d, ids_d)
        return loss_fn(out.embeddings_a, out.embeddings_b)

    S_img = 197
    # (N, K) of the fp32-residual GEMMs: out-projection and MLP-down, any M.  A bracket is two event records on the launch stream (~2 us of
    # pipeline bubble each: 48 per step measured +0.09 ms): the launches of the first 8 timed steps are bracketed, not all of them
    probe = ops.GemmProbe(shapes=[(768, 768), (768, 3072)], max_samples=24 * 8)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    graph, graph_note = None, None
    with torch.no_grad():
        for _ in range(args.warmup):
            loss = step()
        if args.graph:
            # ONE HIP graph for the whole step (the C-ABI neither allocates nor synchronises; the packed all-gather of N > 1 is captured
            # with it): K replays are timed, the loss is read from the graph's output buffer afterwards
            try:
                cap = torch.cuda.Stream(device=dev)
                cap.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(cap):
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=cap):
                        loss = step()
                torch.cuda.current_stream(dev).wait_stream(cap)
                for _ in range(2):
                    graph.replay()
            except Exception as e:  # noqa: BLE001
                graph, graph_note = None, f"capture failed, eager launches timed instead: {type(e).__name__}: {e}"
                torch.cuda.synchronize(dev)
                loss = step()
        fence(dev, dist, 2 if use_dist else 1)
        t0 = time.perf_counter()
        marks[0].record()
        if graph is not None:
            for i in range(args.steps):
                graph.replay()
                marks[i + 1].record()
        elif args.no_probe:
            for i in range(args.steps):
                loss = step()
                marks[i + 1].record()
        else:
            with probe:
                for i in range(args.steps):
                    loss = step()
                    marks[i + 1].record()
        t_enqueued = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        dt_local = time.perf_counter() - t0   # this rank's own K steps, before it waits for the others
        fence(dev, dist, 2 if use_dist else 1)
        dt = time.perf_counter() - t0
        # the loss's ONE collective on its own (N > 1): the packed [B, 2E] all-gather, HIP-event timed on this rank, outside the timed region
        allgather_ms = None
        if use_dist:
            from multimodal_amd.utils.distributed import gather_packed_features

            out = model(images_d, ids_d)
            for _ in range(3):
                gather_packed_features(out.embeddings_a, out.embeddings_b)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                gather_packed_features(out.embeddings_a, out.embeddings_b)
            e1.record()
            torch.cuda.synchronize(dev)
            allgather_ms = e0.elapsed_time(e1) / 20
    share_check = None
    if args.share_gpu and use_dist:
        # the 8c protocol (3) inside the bench's own control flow: mean over ranks of the per-rank losses == the one-process loss on the gathered batch
        from multimodal_amd.utils.distributed import gather_packed_features

        with torch.no_grad():
            out = model(images_d, ids_d)
            buf, r_, w_ = gather_packed_features(out.embeddings_a, out.embeddings_b)
            assert (r_, w_) == (rank, world) and buf.shape[0] == world * B
            E_ = buf.shape[1] // 2
            out3, la_, _ = ops.contrastive_fwd(buf[:, :E_], buf[:, E_:], buf[:, :E_], buf[:, E_:], 2 * E_, loss_fn.logit_scale.detach().reshape(1), 0)
            mine = loss_fn(out.embeddings_a, out.embeddings_b).reshape(1).float()
            dist.all_reduce(mine)
            share_check = {"loss_mean_over_ranks": float(mine) / world, "loss_one_process_on_gathered": float(out3[0]),
                           "logits_block": list(la_.shape)}
    loss_val = float(loss)
    if not math.isfinite(loss_val):
        raise SystemExit(f"non-finite loss {loss_val}")
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = per_step[len(per_step) // 2] if per_step else float("nan")

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    per_rank = torch.tensor([dt_local, t_enqueued, allgather_ms or 0.0], dtype=torch.float64, device=dev)
    per_rank_all = [per_rank]
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per_rank_all = [torch.zeros_like(per_rank) for _ in range(world)]
        dist.all_gather(per_rank_all, per_rank)
    dt_max = float(t)
    pairs_per_s = world * B * args.steps / dt_max

    def family(samples):
        """Σ algorithmic FLOPs / Σ duration of a list of timed launches (each possibly a grouped launch of two problems)."""
        fl = sum(2.0 * m * n * k + (2.0 * c[0] * c[1] * c[2] if c else 0.0) for _, (m, n, k), c in samples)
        ms = sum(d for d, _, _ in samples)
        return fl, ms

    roofline = None
    other_kernels = None
    bf16_resident = None
    samples = [] if args.no_probe else probe.samples()
    if samples:
        with torch.no_grad():
            # secondary kernels (MLP-up + QuickGELU, qkv) timed the same way in 3 extra steps AFTER the timed region
            probe2 = ops.GemmProbe(shapes=[(3072, 768), (2304, 768)])
            with probe2:
                for _ in range(3):
                    step()
            torch.cuda.synchronize(dev)
            # the r03 input form (bf16-resident image batch), 10 steps, for continuity with BENCH_r03
            if not args.bf16_images:
                keep = images_d
                images_d = images_bf16
                for _ in range(3):
                    step()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    step()
                e1.record()
                torch.cuda.synchronize(dev)
                images_d = keep
                bf16_resident = {"ms_per_step": round(e0.elapsed_time(e1) / 10, 3), "steps": 10,
                                 "note": "image batch cast to bf16 once outside the timed steps (the r03 bench default); this rank only"}
        s2 = probe2.samples()
        other_kernels = []
        for (n_, k_), name in (((3072, 768), "grouped MLP-up (+bias, QuickGELU)"), ((2304, 768), "grouped qkv (+bias)")):
            sub = [x for x in s2 if x[1][1:] == (n_, k_)]
            if sub:
                fl, ms = family(sub)
                other_kernels.append({"kernel": f"gemm_bf16_nt_kernel_ppg<false,{1 if n_ == 3072 else 0}>: {name}", "launches_timed": len(sub),
                                      "launch_ms": round(ms / len(sub), 4), "achieved": round(fl / (ms * 1e-3) / 1e12, 2),
                                      "frac": round(fl / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "where": "3 extra steps after the timed region"})
        fl, ms = family(samples)
        achieved = fl / (ms * 1e-3) / 1e12
        by_shape = {}
        for (n_, k_), name in (((768, 768), "out_projection"), ((768, 3072), "mlp_down")):
            sub = [x for x in samples if x[1][1:] == (n_, k_)]
            if sub:
                f2, m2 = family(sub)
                by_shape[name] = {"launches_timed": len(sub), "launch_ms": round(m2 / len(sub), 4), "achieved": round(f2 / (m2 * 1e-3) / 1e12, 2),
                                  "frac": round(f2 / (m2 * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4), "first_problem_MNK": list(sub[0][1]),
                                  "companion_MNK": list(sub[0][2]) if sub[0][2] else None}
        traffic = None
        pmc = ROOT / "profiles" / "pmc_residual_kernel.json"
        if pmc.exists():
            try:
                z = json.loads(pmc.read_text())["shapes"]
                traffic = int(sum(v["hbm_bytes_per_launch"] for v in z.values()) / len(z))
            except Exception:
                traffic = None
        # MEASURED clocks (r05): amdsmi telemetry of 5-s loops of these very launches (socket power / gfxclk at 20 Hz; 1400 W cap) -- a stored figure from
        # profiles/r06_power_clocks_summary.json, not this run.  `frac` stays against the nominal 2.4 GHz peak; `frac_at_measured_clock` is against what
        # the clock the chip sustains under that kernel allows.  (Until r04 this field was derived from GRBM_GUI_ACTIVE, which misreads short kernels.)
        tele = ROOT / "profiles" / "r06_power_clocks_summary.json"
        if tele.exists():
            try:
                tk = json.loads(tele.read_text())["kernels"]
                for name in by_shape:
                    if name in tk:
                        mhz = tk[name]["gfxclk_mhz"]
                        by_shape[name]["telemetry"] = {"gfxclk_mhz": mhz, "socket_power_w": tk[name]["socket_power_w"], "power_cap_w": 1400,
                                                       "source": "profiles/r06_power_clocks.json (stored; amdsmi at 20 Hz over 5-s loops of this launch)"}
                        by_shape[name]["frac_at_measured_clock"] = round(by_shape[name]["achieved"] / (MFMA_BF16_PEAK_TFLOPS * mhz / 2400.0), 4)
            except Exception:
                pass
        roofline = {"bound": "mfma",
                    "kernel": "gemm_bf16_nt_kernel_ppg<true,0>: grouped out-projection / MLP-down of both towers (+bias, fp32 residual read-modify-write); "
                              "the kernel with the largest share of the step",
                    "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic,
                    "traffic_source": None if traffic is None else "profiles/pmc_residual_kernel.json: HBM bytes per launch (mean of the two shapes) from separate "
                                                                    "rocprofv3 --pmc passes (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE); a stored "
                                                                    "figure from whole-chip launches, not measured in this run",
                    "launch_ms": round(ms / len(samples), 4), "launches_timed": len(samples),
                    # (24 launches per step -- 12 layers x 2 shapes -- of which the first 8 timed steps are bracketed)
                    "share_of_step": round(24 * (ms / len(samples)) / (dt_local / args.steps * 1e3), 4),
                    "algorithmic_flops_per_launch": fl / len(samples), "by_shape": by_shape, "other_kernels": other_kernels}

    cpu_baseline = cpu_baseline_leg(sd_host, images, ids, args.cpu_sample, full=bool(args.cpu_full)) if sd_host is not None else None
    from multimodal_amd.schedule import get_schedule

    sch = get_schedule()
    schedule_desc = {"two_tower": sch.two_tower, "side_stream": sch.side_stream}

    if rank == 0:
        line = {
            "metric": METRIC,
            "value": round(pairs_per_s, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3), "ms_per_step_median": round(median_ms, 3),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "CLIP ViT-B/16 + text transformer forward + ContrastiveLossWithTemperature "
                                   f"({'global, packed RCCL all-gather' if use_dist else 'local'}), random-init weights",
                       "per_gpu_batch": B, "global_batch": world * B, "seq_img": S_img, "seq_txt": 77, "image_dtype": "bf16" if args.bf16_images else "fp32",
                       "parallelism": f"dp{world}", "gemm_variant": args.gemm_variant, "schedule": schedule_desc},
            "loss": round(loss_val, 5),
            "step_mfma_frac": round(pairs_per_s / world * GF_PER_PAIR * 1e9 / (MFMA_BF16_PEAK_TFLOPS * 1e12), 4),
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        if bf16_resident is not None:
            line["bf16_resident_images"] = bf16_resident
        # diagnosability of the N > 1 runs: each rank's own time for its K steps (before the closing barrier), its host time to enqueue them,
        # and the all-gather alone; `value` above is computed from the max-over-ranks fenced wall time as the contract says
        line["per_rank_ms_per_step"] = [round(float(x[0]) / args.steps * 1e3, 3) for x in per_rank_all]
        line["per_rank_host_enqueue_ms_per_step"] = [round(float(x[1]) / args.steps * 1e3, 3) for x in per_rank_all]
        line["graph"] = graph is not None
        if graph_note:
            line["graph_note"] = graph_note
        if affinity is not None:
            line["cpu_affinity_rank0"] = affinity
        if share_check is not None:
            line.update(share_check)
            line["share_gpu_test"] = True
            line["ranks_seen"] = ranks_seen
            line["note"] = "TEST MODE (--share-gpu): all ranks on device 0 over gloo; value is not a measurement"
        if ranks_seen is not None:
            line["rccl_ranks_seen"] = None if args.share_gpu else ranks_seen
            line["allgather_ms"] = [round(float(x[2]), 4) for x in per_rank_all]
        os.write(record_fd, (json.dumps(line) + "\n").encode())
    os.close(record_fd)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
