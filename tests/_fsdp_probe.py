"""Helper of tests/test_gpu_fsdp_single_rank.py: ONE rank of a FullyShardedDataParallel job on a single MI355X.

    WORLD_SIZE=1, backend nccl (= RCCL): what a one-GPU box can run of the reference trainer's wrapping (FSDP switches to NO_SHARD at world size 1).
    WORLD_SIZE=2, backend gloo: two processes share the one GPU (two RCCL ranks cannot sit on one device) -> FULL_SHARD for real: every
        wrapped layer's parameters exist only around that layer's own forward / backward.

The model is the reference trainer's (examples/flava/native/train.py:183-206): FLAVA for pre-training (small config, weights and inputs of
the committed fixtures), wrapped with transformer_auto_wrap_policy over {TransformerEncoderLayer, ImageTransformer, BERTTextEncoder,
FLAVATransformerWithoutEmbeddings}.  Every rank feeds the SAME batch, so the gradient average over ranks equals the one-process gradient
and the unwrapped model on this process is the reference for everything: eval outputs before training, two SGD steps (losses), eval
outputs after them (a stale packed-weight cache would show here), the final parameters.  Prints one JSON line."""
import copy
import functools
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


class PreTrain(nn.Module):
    """model + loss heads in one module, like the reference's FLAVAPreTrainModule: forward -> the summed pre-training losses."""

    def __init__(self, model, loss):
        super().__init__()
        self.model, self.loss = model, loss

    def forward(self, image, text, patches_mask, text_masked, itm, mim, mlm):
        out = self.model(image, text, image_patches_mask=patches_mask, text_masked=text_masked)
        lo = self.loss(image_sequence=out.image.last_hidden_state, text_sequence=out.text.last_hidden_state,
                       image_masked_sequence=out.image_masked.last_hidden_state, text_masked_sequence=out.text_masked.last_hidden_state,
                       multimodal_masked_sequence=out.multimodal_masked.last_hidden_state, itm_labels=itm, mim_labels=mim, mlm_labels=mlm,
                       projected_image_embeddings=out.projected_image_embeddings, projected_text_embeddings=out.projected_text_embeddings)
        total = sum(getattr(lo.losses, n) for n in ("itm_loss", "mmm_text_loss", "mmm_image_loss", "global_contrastive_loss"))
        return total, out


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("FSDP_PROBE_BACKEND", "nccl" if world == 1 else "gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from multimodal_amd import build

    build.build()
    from multimodal_amd._autograd import plain_layers
    from multimodal_amd.models.flava.image_encoder import ImageTransformer
    from multimodal_amd.models.flava.model import flava_model
    from multimodal_amd.models.flava.transformer import FLAVATransformerWithoutEmbeddings, TransformerEncoderLayer
    from multimodal_amd.modules.encoders.bert_text_encoder import BERTTextEncoder
    from multimodal_amd.modules.losses.flava import FLAVAPretrainingLoss
    from tests._util import fixture_sd
    from tests.golden.make_golden_flava_grad import SMALL_KW

    g = ROOT / "tests" / "golden"
    z, zl = np.load(g / "flava_small.npz"), np.load(g / "flava_pretrain_small.npz")
    model = flava_model(**SMALL_KW)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in fixture_sd(z).items()}, strict=True)
    loss = FLAVAPretrainingLoss(hidden_size=128, text_vocab_size=200, image_vocab_size=64)
    loss.load_state_dict({k: torch.from_numpy(v) for k, v in fixture_sd(zl).items()}, strict=True)
    plain = PreTrain(model, loss).to(dev)
    T = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
    batch = (T(z["image"]), T(z["text"]), T(z["patches_mask"]), T(z["text_masked"]), T(zl["itm_labels"]), T(zl["mim_labels"]), T(zl["mlm_labels"]))

    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch.distributed.fsdp import FullyShardedDataParallel as FSDP
    from torch.distributed.fsdp.wrap import transformer_auto_wrap_policy

    use_orig = os.environ.get("FSDP_PROBE_USE_ORIG_PARAMS", "0") == "1"
    to_wrap = copy.deepcopy(plain)
    # torch >= 2.1 FSDP refuses 0-dim parameters ("FSDP doesn't support scalar parameters"); the contrastive loss's logit_scale is one in the
    # reference too (modules/losses/flava.py:262, contrastive_loss_with_temperature.py:160): it stays an ordinary replicated parameter
    scalars = [p for p in to_wrap.parameters() if p.dim() == 0]
    res_scalars = len(scalars)
    wrapped = FSDP(to_wrap, device_id=dev, limit_all_gathers=True, use_orig_params=use_orig, ignored_states=scalars,
                   auto_wrap_policy=functools.partial(transformer_auto_wrap_policy, transformer_layer_cls={
                       TransformerEncoderLayer, ImageTransformer, BERTTextEncoder, FLAVATransformerWithoutEmbeddings}))
    res = {"rank": rank, "world": world, "backend": backend, "use_orig_params": use_orig, "sharding": str(wrapped.sharding_strategy),
           "ignored_scalar_params": res_scalars}
    inner = wrapped.module
    enc_layers = inner.model.image_encoder.encoder.layer
    res["layers_are_fsdp"] = all(isinstance(m, FSDP) for m in enc_layers) and isinstance(inner.model.image_encoder, FSDP) and isinstance(
        inner.model.mm_encoder, FSDP)
    res["plain_layers_sees_wrapping"] = not plain_layers(enc_layers, TransformerEncoderLayer)
    res["n_fsdp_units"] = sum(1 for m in wrapped.modules() if isinstance(m, FSDP))

    def eval_out(m):
        m.eval()
        with torch.no_grad():
            total, out = m(*batch)
        m.train()
        return [float(total), out.projected_image_embeddings.float().clone(), out.multimodal_masked.last_hidden_state.float().clone(),
                out.image.attentions[0].float().clone() if out.image.attentions else None]

    def close(a, b, tol):
        if a is None or b is None:
            return 0.0 if a is b else float("inf")
        return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12) / tol

    sd0 = {k: v.detach().clone() for k, v in plain.state_dict().items()}
    e0p, e0w = eval_out(plain), eval_out(wrapped)
    res["eval_before_dloss"] = abs(e0p[0] - e0w[0])
    res["eval_before_worst_rel_over_tol"] = max(close(a, b, 1e-3) for a, b in zip(e0w[1:], e0p[1:]))
    losses = {"plain": [], "fsdp": []}
    for name, m in (("plain", plain), ("fsdp", wrapped)):
        m.train()
        opt = torch.optim.SGD(m.parameters(), lr=0.05)
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            total, _ = m(*batch)
            total.backward()
            opt.step()
            losses[name].append(float(total))
    res["losses"] = losses
    res["loss_moved"] = abs(losses["plain"][1] - losses["plain"][0])
    res["dloss_steps"] = [abs(a - b) for a, b in zip(losses["plain"], losses["fsdp"])]
    e1p, e1w = eval_out(plain), eval_out(wrapped)
    res["eval_after_dloss"] = abs(e1p[0] - e1w[0])
    # (4e-3 = one bf16 ulp (2^-8) of the tensor's largest value: the two arrangements round at different places -- per-layer nodes against the stack-level
    #  node, fp32 atomics in the embedding gradients, and since r05 a bf16 gradient handed to each LayerNorm backward -- and two lr = 0.05 steps carry that
    #  into the outputs, where the bf16 forward turns any parameter difference into whole-ulp flips; measured 2.2e-3 on the two-rank FULL_SHARD case,
    #  < 2e-3 on one rank)
    res["eval_after_worst_rel_over_tol"] = max(close(a, b, 4e-3) for a, b in zip(e1w[1:], e1p[1:]))
    res["eval_changed_by_training"] = abs(e1p[0] - e0p[0])
    # final parameters: FSDP's full state dict (clean names; gathers the shards) against the unwrapped model's
    sd_w = wrapped.state_dict()
    sd_p = plain.state_dict()
    res["state_dict_keys_equal"] = sorted(sd_w.keys()) == sorted(sd_p.keys())
    # The UPDATES the two steps made to every parameter (final - initial), wrapped vs unwrapped: relative to the tensor's largest update, with a
    # floor at 1e-3 of the largest update of any tensor -- parameters whose gradient is mathematically zero (the attention key biases: softmax is
    # invariant to a per-query shift) move by round-off on both sides, where a relative figure means nothing.  What differs between the two runs is
    # the order of fp32 atomics (embedding gradients) and one autograd node per layer instead of one per stack: the same kernels on the same
    # values, so the updates agree far inside the bf16-gradient tolerance of the gradient fixtures (6e-2).
    upd_max = max(float((sd_p[k].float() - sd0[k].float()).abs().max()) for k in sd_p if sd_p[k].dtype.is_floating_point)
    worst, worst_k, worst_abs = 0.0, "", 0.0
    for k, v in sd_p.items():
        if k in sd_w and v.dtype.is_floating_point:
            up, uw = v.float() - sd0[k].float(), sd_w[k].to(v.device).float() - sd0[k].float()
            d = float((uw - up).abs().max()) / max(float(up.abs().max()), 1e-3 * upd_max)
            if d > worst:
                worst, worst_k, worst_abs = d, k, float((uw - up).abs().max())
    res["params_worst_rel"] = worst
    res["params_worst_key"] = worst_k
    res["params_worst_abs"] = worst_abs
    res["largest_update"] = upd_max
    print("FSDP_PROBE_RESULT " + json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
