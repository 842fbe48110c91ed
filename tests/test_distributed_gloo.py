"""N>1 path on CPU: world_size-2 (and 4) gloo process groups.  Covers the host logic the 8-GPU run depends on —
gather_tensor semantics (reference tests/utils/test_distributed.py:37-59), the packed single-collective gather's
layout, rank label offsets — with the per-rank arithmetic checked by the oracle against the reference's own
gloo outputs (tests/golden/loss_dist.npz)."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
GOLDEN = ROOT / "tests" / "golden"


def _init(rank, world, sync):
    import torch.distributed as dist

    if str(ROOT) not in sys.path:
        sys.path.insert(0, str(ROOT))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"file://{sync}", world_size=world, rank=rank)
    return dist


def _gather_worker(rank, world, sync, base, bp_name):
    dist = _init(rank, world, sync)
    from multimodal_amd.utils.distributed import BackpropType, concat_gather_all_gpu, gather_tensor, get_rank

    bp = BackpropType[bp_name]
    t = base.clone().requires_grad_() + rank
    out = gather_tensor(t, bp)
    assert len(out) == world and get_rank() == rank
    for i, g in enumerate(out):
        assert torch.allclose(g, base + i)
        if (bp == BackpropType.LOCAL and i == rank) or bp == BackpropType.GLOBAL:
            assert g.grad_fn is not None
        else:
            assert g.grad_fn is None
    cat = concat_gather_all_gpu(t, bp)
    assert cat.shape == (world * base.shape[0], base.shape[1])
    dist.destroy_process_group()


@pytest.mark.parametrize("bp", ["GLOBAL", "LOCAL", "NONE"])
def test_gather_tensor_world2(tmp_path, bp):
    torch.manual_seed(1234)
    mp.spawn(_gather_worker, (2, str(tmp_path / "sync"), torch.randn(4, 8), bp), nprocs=2)


def _loss_worker(rank, world, sync):
    dist = _init(rank, world, sync)
    from multimodal_amd.utils.distributed import gather_packed_features
    from oracle import clip_oracle as oc

    z = np.load(GOLDEN / "loss_dist.npz")
    a_all, b_all = torch.from_numpy(z["a_all"]), torch.from_numpy(z["b_all"])
    B, E = a_all.shape[0] // world, a_all.shape[1]
    a, b = a_all[rank * B:(rank + 1) * B].contiguous(), b_all[rank * B:(rank + 1) * B].contiguous()
    buf, r, w = gather_packed_features(a, b)
    assert (r, w) == (rank, world) and buf.shape == (world * B, 2 * E)
    # layout contract of the loss kernel: [:, :E] = every rank's a in rank order, [:, E:] = b
    assert torch.equal(buf[:, :E], a_all) and torch.equal(buf[:, E:], b_all)
    o = oc.contrastive_loss_with_temperature(a.numpy(), b.numpy(), np.log(1 / 0.07), buf[:, :E].numpy(), buf[:, E:].numpy(), rank=r)
    pre = f"w{world}.GLOBAL.r{rank}."
    assert abs(float(o["loss"]) - float(z[pre + "loss"])) < 2e-6
    assert np.abs(o["logits_a"] - z[pre + "logits_a"]).max() < 2e-5
    losses = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(losses, torch.tensor([float(o["loss"])]))
    single = oc.contrastive_loss_with_temperature(a_all.numpy(), b_all.numpy(), np.log(1 / 0.07))
    assert abs(float(torch.stack(losses).mean()) - float(single["loss"])) < 1e-5
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_packed_gather_and_rank_offsets(tmp_path, world):
    mp.spawn(_loss_worker, (world, str(tmp_path / "sync")), nprocs=world)


def test_packed_gather_without_process_group():
    from multimodal_amd.utils.distributed import gather_packed_features

    a, b = torch.randn(3, 4), torch.randn(3, 4)
    buf, r, w = gather_packed_features(a, b)
    assert (r, w) == (0, 1) and torch.equal(buf[:, :4], a) and torch.equal(buf[:, 4:], b)
    with pytest.raises(ValueError):
        gather_packed_features(a, torch.randn(2, 4))


def test_packed_gather_recognises_the_two_halves_of_one_block():
    """CLIP.forward normalises both towers into ONE [B, 2E] buffer and returns its halves: the gather must send that buffer as is (same
    storage, no packing copies) and anything else through the packing path."""
    from multimodal_amd.utils.distributed import gather_packed_features

    block = torch.randn(5, 12)
    a, b = block[:, :6], block[:, 6:]
    buf, r, w = gather_packed_features(a, b)
    assert (r, w) == (0, 1) and buf.data_ptr() == block.data_ptr() and torch.equal(buf, block)
    buf2, _, _ = gather_packed_features(b, a)  # swapped halves are NOT the packed layout: copied into the right order
    assert buf2.data_ptr() != block.data_ptr() and torch.equal(buf2[:, :6], b) and torch.equal(buf2[:, 6:], a)
    wide = torch.randn(5, 20)
    buf3, _, _ = gather_packed_features(wide[:, :6], wide[:, 6:12])  # row stride 20 != 2E
    assert buf3.shape == (5, 12) and torch.equal(buf3[:, 6:], wide[:, 6:12])
