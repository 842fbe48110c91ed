"""FullyShardedDataParallel wrapping of the drop-in FLAVA modules, exactly as the reference's trainer wraps them
(/root/reference/examples/flava/native/train.py:183-206: transformer_auto_wrap_policy over TransformerEncoderLayer, ImageTransformer,
BERTTextEncoder, FLAVATransformerWithoutEmbeddings; SURVEY.md section 2.2: "FSDP/DDP wrapping still works" is part of the module contract;
VERDICT r04 missing 3).

What FSDP changes under the modules: a wrapped layer's parameters are plain Tensor VIEWS of the unit's gathered flat parameter and exist
only around that layer's own forward / backward, so the stack-level autograd nodes and grouped launches (which read all layers' parameters
at once) cannot be used: the stacks detect wrapped / hooked layers (_autograd.plain_layers) and CALL them one by one, each layer running its
own one-layer autograd node; the packed-weight caches do not cache such views (_packing._cacheable).

Each case runs tests/_fsdp_probe.py in its own process(es): eval outputs before training, two SGD steps, eval outputs after them, final
parameters -- all against the unwrapped model in the same process."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _run(world: int, extra_env=None):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "_fsdp_probe.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    results = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, out[-4000:]
        line = [ln for ln in out.splitlines() if ln.startswith("FSDP_PROBE_RESULT ")]
        assert line, out[-4000:]
        results.append(json.loads(line[-1][len("FSDP_PROBE_RESULT "):]))
    return results


def _check(r, sharded: bool):
    assert r["layers_are_fsdp"] and r["plain_layers_sees_wrapping"], r
    assert r["n_fsdp_units"] == 1 + 3 + 6, r            # root + the three encoders + 2 layers in each of them
    assert ("FULL_SHARD" in r["sharding"]) == sharded, r
    assert r["state_dict_keys_equal"], r
    # inference through the wrapped model == the unwrapped model (same kernels; per-layer launches instead of grouped ones)
    assert r["eval_before_dloss"] <= 2e-3 and r["eval_before_worst_rel_over_tol"] <= 1.0, r
    # two training steps: the same losses (per-layer autograd nodes == the stack-level node's arithmetic; embedding gradients use fp32 atomics)
    assert r["loss_moved"] > 1e-3, r                      # the steps did change the model
    assert max(r["dloss_steps"]) <= 5e-3, r
    # after training: eval forwards of the wrapped model use the UPDATED parameters (no stale packed copies) and match the unwrapped model
    assert r["eval_changed_by_training"] > 1e-3, r
    assert r["eval_after_dloss"] <= 5e-3 and r["eval_after_worst_rel_over_tol"] <= 1.0, r
    # the two steps moved every parameter by the same amount (relative to the tensor's largest update; tests/_fsdp_probe.py)
    assert r["params_worst_rel"] <= 3e-2, (r["params_worst_key"], r["params_worst_rel"], r["params_worst_abs"], r["largest_update"])


def test_flava_pretraining_under_fsdp_one_rccl_rank():
    """The reference trainer's wrapping on one RCCL rank (FSDP turns into NO_SHARD at world size 1): flat parameters, view parameters inside
    the layers, per-layer autograd nodes, optimizer on the flat parameters."""
    (r,) = _run(1)
    assert r["backend"] == "nccl"
    _check(r, sharded=False)


def test_flava_pretraining_under_fsdp_one_rccl_rank_use_orig_params():
    (r,) = _run(1, {"FSDP_PROBE_USE_ORIG_PARAMS": "1"})
    _check(r, sharded=False)


def test_flava_pretraining_under_fsdp_full_shard_two_ranks_on_one_gpu():
    """FULL_SHARD for real: two processes share the GPU over gloo; a layer's parameters are gathered only around its own forward and
    backward, the rest of the time every rank holds half of each flat parameter."""
    rs = _run(2)
    for rank, r in enumerate(rs):
        assert r["rank"] == rank and r["world"] == 2
        _check(r, sharded=True)


def test_fsdp_wrapped_training_step_price_is_bounded():
    """What the reference trainer's wrapping costs at cfg 4's real size (FLAVA B = 128, forward + pre-training loss + backward + SGD; one RCCL
    rank, NO_SHARD, transformer_auto_wrap_policy over the encoder layers and the three encoders: tools/flava_bench.py --train --fsdp).
    Measured r06 (profiles/r06_fsdp_price.txt, r06_fsdp_prof.txt): 83.5 ms unwrapped, 110.7 ms wrapped (+33 %; use_orig_params=True +28 %) with
    the SAME kernels and the same kernel time per step (92.6 / 94.0 ms) -- the price is the lost co-running of the two towers' streams and host
    time in FSDP's per-unit hooks (device idle 13 % -> 21 %), not other kernels.  The bound catches a regression of the wrapped path (e.g. the
    per-layer nodes falling off the HIP kernels), not box-to-box noise."""
    import json as _json
    import subprocess as _sp

    def run(*extra):
        p = _sp.run([sys.executable, str(ROOT / "tools" / "flava_bench.py"), "--train", "--steps", "4", "--warmup", "2", *extra], capture_output=True,
                    text=True, timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert p.returncode == 0 and lines, (p.stdout + p.stderr)[-3000:]
        return _json.loads(lines[-1])

    plain, wrapped = run(), run("--fsdp")
    assert wrapped["fsdp"]["units"] == 34 and "NO_SHARD" in wrapped["fsdp"]["sharding"]
    assert abs(wrapped["last"] - plain["last"]) <= 2e-2 * max(1.0, abs(plain["last"]))  # the same training trajectory (loss after 6 steps)
    ratio = wrapped["ms_per_step"] / plain["ms_per_step"]
    print(f"FSDP price: {plain['ms_per_step']:.1f} -> {wrapped['ms_per_step']:.1f} ms per step (x{ratio:.2f})")
    assert ratio <= 1.6, (plain["ms_per_step"], wrapped["ms_per_step"])
