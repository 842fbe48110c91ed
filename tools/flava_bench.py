"""FLAVA forward (+ pre-training loss) timing on one MI355X — SURVEY.md section 8 cfg 4 (B = 128), WITHOUT the DALL-E
codebook stage (out of scope, section 8f rank 2):  python tools/flava_bench.py [--batch 128] [--steps 10]

Algorithmic FLOPs per sample: image 35.13 GF and text 13.30 GF per pass, fusion 24.75 GF, heads ~1.8 GF.
The reference runs 2 image + 2 text passes; so does this path when an image_patches_mask is given.
Prints one JSON line."""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-loss", action="store_true")
    ap.add_argument("--train", action="store_true", help="training step: forward + pre-training loss + backward + SGD")
    ap.add_argument("--two-pass-train", action="store_true", help="A/B arm: schedule.flava_batched_train = False -- the unmasked and the masked pass of a tower as the reference's two calls in training (default: one 2B pass)")
    ap.add_argument("--grouped", action="store_true", help="schedule.flava_grouped: image and text towers layer-locked with grouped launches")
    ap.add_argument("--no-attentions", action="store_true", help="schedule.flava_attentions = False: the forwards do not produce the attention probabilities (opt-out)")
    ap.add_argument("--probs-two-pass", action="store_true", help="A/B: unmasked attention probabilities from the two-pass kernel (debug variant 514) instead of flash + one pass")
    ap.add_argument("--copy-trace", action="store_true", help="after the timing: one more step under the torch profiler -> which host call sites issue copies / fills / cats / casts")
    ap.add_argument("--fsdp", action="store_true", help="with --train: model + loss heads wrapped by FullyShardedDataParallel exactly as the reference trainer does "
                    "(examples/flava/native/train.py:183-206: transformer_auto_wrap_policy over the encoder layers and the three encoders), one RCCL rank (NO_SHARD)")
    ap.add_argument("--fsdp-orig-params", action="store_true", help="use_orig_params=True for --fsdp")
    ap.add_argument("--codebook", action="store_true", help="MIM labels from the DALL-E codebook (112x112 images) inside the step instead of synthetic ones")
    a = ap.parse_args()
    from multimodal_amd.models.flava.model import flava_model
    from multimodal_amd.modules.losses.flava import FLAVAPretrainingLoss

    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    if a.probs_two_pass:
        from multimodal_amd import _lib

        _lib.lib().mmamd_debug_set_attn_variant(514)
    if a.no_attentions:
        from multimodal_amd.schedule import set_schedule

        set_schedule(flava_attentions=False)
    if a.grouped:
        from multimodal_amd.schedule import set_schedule

        set_schedule(flava_grouped=True)
    if a.two_pass_train:
        from multimodal_amd.schedule import set_schedule

        set_schedule(flava_batched_train=False)
    model = flava_model().to(dev)
    loss = FLAVAPretrainingLoss().to(dev)
    model, loss = (model.train(), loss.train()) if a.train else (model.eval(), loss.eval())
    wrapped, fsdp_units = None, 0
    if a.fsdp:
        assert a.train, "--fsdp prices the wrapped TRAINING step"
        import functools
        import os
        import socket

        import torch.distributed as dist
        from torch import nn
        from torch.distributed.fsdp import FullyShardedDataParallel as FSDP
        from torch.distributed.fsdp.wrap import transformer_auto_wrap_policy

        from multimodal_amd.models.flava.image_encoder import ImageTransformer
        from multimodal_amd.models.flava.transformer import FLAVATransformerWithoutEmbeddings, TransformerEncoderLayer
        from multimodal_amd.modules.encoders.bert_text_encoder import BERTTextEncoder

        class PreTrain(nn.Module):  # the reference's FLAVAPreTrainModule shape: model + loss heads in ONE module, which the trainer wraps
            def __init__(self, model, loss):
                super().__init__()
                self.model, self.loss = model, loss

            def forward(self, image, text, pmask, text_masked, itm, mim, mlm):
                o = self.model(image, text, image_patches_mask=pmask, text_masked=text_masked)
                lo = self.loss(image_sequence=o.image.last_hidden_state, text_sequence=o.text.last_hidden_state,
                               image_masked_sequence=o.image_masked.last_hidden_state, text_masked_sequence=o.text_masked.last_hidden_state,
                               multimodal_masked_sequence=o.multimodal_masked.last_hidden_state, itm_labels=itm, mim_labels=mim, mlm_labels=mlm,
                               projected_image_embeddings=o.projected_image_embeddings, projected_text_embeddings=o.projected_text_embeddings)
                return lo.losses.itm_loss + lo.losses.mmm_text_loss + lo.losses.mmm_image_loss + lo.losses.global_contrastive_loss

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        both = PreTrain(model, loss)
        scalars = [p for p in both.parameters() if p.dim() == 0]  # FSDP refuses 0-dim parameters (logit_scale): replicated, as in the probe
        wrapped = FSDP(both, device_id=dev, limit_all_gathers=True, use_orig_params=a.fsdp_orig_params, ignored_states=scalars,
                       auto_wrap_policy=functools.partial(transformer_auto_wrap_policy, transformer_layer_cls={
                           TransformerEncoderLayer, ImageTransformer, BERTTextEncoder, FLAVATransformerWithoutEmbeddings}))
        fsdp_units = sum(1 for m in wrapped.modules() if isinstance(m, FSDP))
        wrapped.train()
    opt = (torch.optim.SGD(wrapped.parameters() if wrapped is not None else list(model.parameters()) + list(loss.parameters()), lr=1e-4)) if a.train else None
    B = a.batch
    g = torch.Generator().manual_seed(1)
    image = torch.randn(B, 3, 224, 224, generator=g).to(dev)
    text = torch.randint(1, 30522, (B, 77), generator=g)
    text[:, 60:] = 0
    text_masked = text.clone()
    mlm = torch.full((B, 77), -1, dtype=torch.long)
    sel = torch.rand(B, 77, generator=g) < 0.15
    sel[:, 60:] = False
    mlm[sel] = text[sel]
    text_masked[sel] = 103
    pmask = (torch.rand(B, 196, generator=g) < 0.4).long()
    mim = torch.randint(0, 8192, (B, 196), generator=g)
    mim[pmask == 0] = -1
    itm = torch.ones(B, dtype=torch.long)
    text, text_masked, mlm, pmask, mim, itm = (t.to(dev) for t in (text, text_masked, mlm, pmask, mim, itm))
    vae, img112, keep = None, None, None
    if a.codebook:
        from multimodal_amd import ops
        from multimodal_amd.models.flava.model import DalleVAEEncoder

        vae = DalleVAEEncoder(pretrained=False).to(dev).eval()
        img112 = torch.randn(B, 3, 112, 112, generator=g).to(dev)
        keep = pmask.to(torch.uint8).contiguous()

    def labels():
        if vae is None:
            return mim
        with torch.no_grad():
            return ops.mask_labels_(vae(img112).flatten(1).contiguous(), keep, -1)  # FLAVAForPreTraining.forward :338-343

    def train_step():
        opt.zero_grad(set_to_none=True)
        if wrapped is not None:
            total = wrapped(image, text, pmask, text_masked, itm, labels(), mlm)
            total.backward()
            opt.step()
            return total.detach()
        o = model(image, text, image_patches_mask=pmask, text_masked=text_masked)
        lo = loss(image_sequence=o.image.last_hidden_state, text_sequence=o.text.last_hidden_state,
                  image_masked_sequence=o.image_masked.last_hidden_state, text_masked_sequence=o.text_masked.last_hidden_state,
                  multimodal_masked_sequence=o.multimodal_masked.last_hidden_state, itm_labels=itm, mim_labels=labels(), mlm_labels=mlm,
                  projected_image_embeddings=o.projected_image_embeddings, projected_text_embeddings=o.projected_text_embeddings)
        total = lo.losses.itm_loss + lo.losses.mmm_text_loss + lo.losses.mmm_image_loss + lo.losses.global_contrastive_loss
        total.backward()
        opt.step()
        return total.detach()

    def step():
        if a.train:
            return train_step()
        with torch.no_grad():
            o = model(image, text, image_patches_mask=pmask, text_masked=text_masked)
            if a.no_loss:
                return o.projected_image_embeddings
            lo = loss(image_sequence=o.image.last_hidden_state, text_sequence=o.text.last_hidden_state,
                      image_masked_sequence=o.image_masked.last_hidden_state, text_masked_sequence=o.text_masked.last_hidden_state,
                      multimodal_masked_sequence=o.multimodal_masked.last_hidden_state, itm_labels=itm, mim_labels=labels(), mlm_labels=mlm,
                      projected_image_embeddings=o.projected_image_embeddings, projected_text_embeddings=o.projected_text_embeddings)
            return lo.losses.global_contrastive_loss

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
    gf = (2 * 35.13 + 2 * 13.30 + 24.75 + (0.0 if a.no_loss else 1.8)) * (3 if a.train else 1)
    if a.codebook:
        from tools.codebook_bench import encoder_gflop

        gf += encoder_gflop(vae)  # forward only: the codebook supplies labels
    if a.copy_trace:
        from collections import Counter

        from torch.profiler import ProfilerActivity, profile

        with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
            step()
            torch.cuda.synchronize()
        c = Counter()
        for ev in prof.events():
            if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::zeros", "aten::zero_", "aten::fill_", "aten::cat", "aten::to", "aten::_to_copy",
                           "aten::index", "aten::index_select", "aten::masked_fill_", "aten::where", "aten::add", "aten::mul", "aten::sum", "aten::mean", "aten::stack"):
                frames = [f for f in (ev.stack or []) if "multimodal_amd" in f or "tools/" in f]
                c[(ev.name, str(ev.input_shapes)[:70], (frames[0] if frames else "?")[-100:])] += 1
        for (name, shp, fr), n in c.most_common(60):
            print(f"{n:4d} {name:18s} {shp:70s} {fr}", file=sys.stderr)
    # the arm ran the probability kernels it is named after (launch counters of the library: no profiler needed)
    from multimodal_amd import ops as _ops

    launches = {k: _ops.launch_count(k) for k in ("attention_probs_lse", "attention_probs_fwd")}
    assert a.train or (launches["attention_probs_lse"] > 0) == (not a.probs_two_pass and not a.no_attentions), launches
    print(json.dumps({"workload": ("TRAINING step: " if a.train else "") + "flava_model() fwd" + ("" if a.no_loss else " + FLAVAPretrainingLoss") + (" + bwd + SGD" if a.train else "") + (" + DALL-E codebook labels" if a.codebook else " (no codebook)"),
                      "batch": B, "ms_per_step": round(ms, 3), "samples_per_s": round(B / ms * 1e3, 1), "gflop_per_sample": gf,
                      "tflops": round(B * gf / ms, 1), "mfma_frac": round(B * gf / ms / 2500.0, 4),
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "fsdp": ({"units": fsdp_units, "use_orig_params": a.fsdp_orig_params, "sharding": str(wrapped.sharding_strategy), "ranks": 1} if wrapped is not None else None),
                      "probs_path": "two_pass" if a.probs_two_pass else "flash+one_pass", "launches": launches, "last": float(r.flatten()[0])}))

    if a.fsdp:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
