"""bench.py as the driver calls it: `python bench.py --gpus N ...` with no launcher around it must start its own N ranks.
CPU coverage of that path (the reference's 2-worker pattern: tests/modules/losses/test_contrastive_loss_with_temperature.py:129-199,
tests/test_utils.py:31-42 — mp.spawn + rendezvous; here torch.distributed.run + gloo):
  * `--gpus 2 --backend gloo --dry-run` goes through the self-launch, the rendezvous on 127.0.0.1, the ranks-seen all-reduce, the
    packed all-gather with its layout check, the fenced max-over-ranks timing and prints ONE JSON line from rank 0;
  * `--gpus 2` on a box without GPUs fails at DEVICE SELECTION inside the ranks, not at argument checking."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


def run_bench(*argv, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "1"
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], cwd=str(ROOT), env=env, capture_output=True, text=True,
                          timeout=timeout)


def json_lines(text):
    out = []
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                out.append(json.loads(ln))
            except ValueError:
                pass
    return out


def test_self_launch_two_ranks_gloo_dry_run():
    p = run_bench("--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "3", "--warmup", "1", "--batch", "8")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = json_lines(p.stdout)
    assert len(lines) == 1, p.stdout  # rank 0 only
    line = lines[0]
    assert line["dry_run"] is True and line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["backend"] == "gloo"
    assert line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 16 and line["config"]["parallelism"] == "dp2"
    assert line["ms_per_step"] > 0


def test_single_process_dry_run_needs_no_launcher():
    p = run_bench("--gpus", "1", "--backend", "gloo", "--dry-run", "--steps", "2", "--warmup", "0", "--batch", "4")
    assert p.returncode == 0, p.stderr[-2000:]
    (line,) = json_lines(p.stdout)
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the failure mode of a box without GPUs")
def test_gpus_2_without_devices_fails_at_device_selection():
    p = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert p.returncode != 0
    err = p.stderr + p.stdout
    assert "no HIP device visible" in err  # reached the ranks' device selection ...
    assert "must be launched with" not in err  # ... and did not stop at argument checking


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run", "--backend", "gloo"], cwd=str(ROOT), env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=4" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_force_dist_runs_the_rccl_path_with_one_rank():
    """`--force-dist`: the N > 1 code path of bench.py on ONE MI355X — RCCL communicator, ranks-seen all-reduce, the packed
    all_gather_into_tensor of the [B, 2E] block, rank-offset labels, barrier fences, max over ranks.  Same loss as the local run."""
    lines = {}
    for tag, extra in (("local", []), ("rccl", ["--force-dist"])):
        p = run_bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "64", "--cpu-sample", "0", "--no-probe", *extra)
        assert p.returncode == 0, p.stderr[-2000:]
        out = p.stdout.strip().splitlines()
        assert len(out) == 1, out  # stdout is the record and nothing else (RCCL's own chatter must not land there)
        lines[tag] = json.loads(out[0])
    assert lines["rccl"]["rccl_ranks_seen"] == 1 and "rccl_ranks_seen" not in lines["local"]
    assert "RCCL all-gather" in lines["rccl"]["config"]["workload"]
    assert lines["rccl"]["loss"] == lines["local"]["loss"]


def test_self_launch_eight_ranks_gloo_dry_run():
    """W = 8 — the size of the SCALE run: eight self-launched ranks through the same control flow (rendezvous, ranks-seen all-reduce, packed
    gather with its layout check, per-rank CPU pinning, per-rank times gathered to rank 0, ONE JSON line)."""
    p = run_bench("--gpus", "8", "--backend", "gloo", "--dry-run", "--steps", "2", "--warmup", "1", "--batch", "4", timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    (line,) = json_lines(p.stdout)
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["config"]["global_batch"] == 32 and line["config"]["parallelism"] == "dp8"
    assert len(line["per_rank_ms_per_step"]) == 8 and all(x > 0 for x in line["per_rank_ms_per_step"])
    assert max(line["per_rank_ms_per_step"]) <= line["ms_per_step"] * 1.5 + 1.0
    aff = line["cpu_affinity_rank0"]
    assert aff is None or "error" in aff or aff["count"] >= 1


def test_cpu_pinning_helper_splits_the_visible_cpus():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    before = os.sched_getaffinity(0)
    try:
        got = b.pin_rank_to_local_cpus(1, 2)
        assert got["count"] >= 1 and set(os.sched_getaffinity(0)) <= set(before)
        if len(before) >= 8:
            assert len(os.sched_getaffinity(0)) == len(before) // 2
        os.sched_setaffinity(0, before)
        if len(before) < 32:  # fewer than 4 cores per rank at W = 8: no private slices, the ranks share the (NUMA-local) list
            got8 = b.pin_rank_to_local_cpus(3, 8)
            assert "shared" in got8["policy"] and set(os.sched_getaffinity(0)) == set(before)
    finally:
        os.sched_setaffinity(0, before)


def test_stored_pmc_pass_has_what_the_bench_line_reads():
    """bench.py fills roofline.traffic and the per-shape clock fields from profiles/pmc_residual_kernel.json (separate rocprofv3 --pmc passes, tools/gpu_pmc_residual.sh):
    the file must keep the two shapes with their HBM bytes and effective clocks, or the line silently loses those fields."""
    import json
    from pathlib import Path

    z = json.loads((Path(__file__).resolve().parents[1] / "profiles" / "pmc_residual_kernel.json").read_text())["shapes"]
    assert set(z) >= {"outproj", "mlpdown"}
    for name, v in z.items():
        assert v["hbm_bytes_per_launch"] > v["algorithmic_bytes_per_launch"] * 0.9, name   # traffic cannot be (much) below the algorithmic bytes
        assert 1.0 < v["effective_clock_ghz"] <= 2.45, name                                  # GRBM_GUI_ACTIVE / 8 / duration against the 2.4 GHz nominal clock
        assert 0.0 < v["mfma_busy_frac"] < 1.0 and 0.0 < v["tcc_hit_rate"] < 1.0, name
