"""CLIP ViT-B/16 TRAINING step on one MI355X (forward + loss + backward + SGD step), B = 256 synthetic pairs:
    python tools/train_bench.py [--batch 256] [--steps 5]
Algorithmic FLOPs = 3 x the forward's 41.09 GF/pair (backward = dgrad + wgrad of every GEMM + attention backward at 2.5x).
Prints one JSON line."""
import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--attn-variant", type=int, default=0, help="mmamd_debug_set_attn_variant code (4000 = the two-kernel attention backward, 4001 = single pass, 4002 = fused two-role)")
    ap.add_argument("--gemm-gm", type=int, default=0, help="mmamd_debug_set_gemm_knob(0, gm): tile-order group (8 = the r03 order)")
    ap.add_argument("--no-deferred-ln-reduce", action="store_true", help="A/B arm: every LayerNorm backward reduces its own dgamma / dbeta / column-sum partials (53 small launches per step) instead of one batched launch per stack")
    ap.add_argument("--no-fused-bias", action="store_true", help="A/B arm: bias gradients from the column-sum passes (the r04 form) instead of the wgrad GEMM's own pass")
    ap.add_argument("--no-grouped-wgrad", action="store_true", help="A/B arm: one split-K launch + reduce per Linear's weight gradient instead of one grouped launch per layer")
    ap.add_argument("--tower", choices=("both", "image", "text"), default="both", help="image / text: the training step of ONE tower alone (loss = mean of its embedding's squares): what each tower costs by itself")
    ap.add_argument("--fsdp", action="store_true", help="model + loss in one module wrapped by FullyShardedDataParallel (one unit per tower + the root), one RCCL rank (NO_SHARD): "
                    "what the wrapper costs the step (the towers' stacks are then called through FSDP's pre/post-forward hooks)")
    ap.add_argument("--fsdp-orig-params", action="store_true")
    ap.add_argument("--f32-dh", action="store_true", help="A/B arm: the dgrad GEMMs in front of a LayerNorm backward write fp32 (the form before r05's LayerNorm-backward register fix) instead of bf16")
    a = ap.parse_args()
    from multimodal_amd import _autograd, _lib

    _autograd._BF16_DH = not a.f32_dh
    _autograd._GROUPED_WGRAD = not a.no_grouped_wgrad

    _autograd._FUSED_BIAS_GRAD = not a.no_fused_bias
    _autograd._DEFER_LN_REDUCE = not a.no_deferred_ln_reduce

    _lib.lib().mmamd_debug_set_attn_variant(a.attn_variant)
    _lib.lib().mmamd_debug_set_gemm_knob(0, a.gemm_gm)
    from multimodal_amd.models.clip import clip_vit_b16
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.synthetic import clip_batch

    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    model = clip_vit_b16().to(dev).train()
    loss_fn = ContrastiveLossWithTemperature().to(dev)
    wrapped, fsdp_units = None, 0
    if a.fsdp:
        assert a.tower == "both"
        import functools
        import os
        import socket

        import torch.distributed as dist
        from torch import nn
        from torch.distributed.fsdp import FullyShardedDataParallel as FSDP
        from torch.distributed.fsdp.wrap import transformer_auto_wrap_policy

        from multimodal_amd.models.clip.image_encoder import CLIPViTEncoder
        from multimodal_amd.models.clip.text_encoder import CLIPTextEncoder

        class Step(nn.Module):
            def __init__(self, model, loss_fn):
                super().__init__()
                self.model, self.loss_fn = model, loss_fn

            def forward(self, images, ids):
                out = self.model(images, ids)
                return self.loss_fn(out.embeddings_a, out.embeddings_b)

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        both = Step(model, loss_fn)
        scalars = [p for p in both.parameters() if p.dim() == 0]
        wrapped = FSDP(both, device_id=dev, limit_all_gathers=True, use_orig_params=a.fsdp_orig_params, ignored_states=scalars,
                       auto_wrap_policy=functools.partial(transformer_auto_wrap_policy, transformer_layer_cls={CLIPViTEncoder, CLIPTextEncoder}))
        fsdp_units = sum(1 for m in wrapped.modules() if isinstance(m, FSDP))
        wrapped.train()
    opt = torch.optim.SGD(wrapped.parameters() if wrapped is not None else list(model.parameters()) + list(loss_fn.parameters()), lr=1e-4)
    images, ids = clip_batch(a.batch)
    images, ids = images.to(dev), ids.to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        if wrapped is not None:
            loss = wrapped(images, ids)
        elif a.tower == "both":
            out = model(images, ids)
            loss = loss_fn(out.embeddings_a, out.embeddings_b)
        else:
            emb = model.encoder_a(images) if a.tower == "image" else model.encoder_b(ids)
            loss = (emb.float() ** 2).mean()
        loss.backward()
        opt.step()
        return loss

    losses = []
    for _ in range(a.warmup):
        losses.append(float(step()))
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(a.steps):
        loss = step()
    t1.record()
    torch.cuda.synchronize()
    losses.append(float(loss))
    ms = t0.elapsed_time(t1) / a.steps
    gf = 3 * 41.09
    from multimodal_amd import ops as _ops

    launches = {k: _ops.launch_count(k) // (a.steps + a.warmup) for k in ("colsum_stage2_batched", "colsum", "gemm_bf16_splitk", "gemm_bf16_tn_splitk_group", "layernorm_bwd")}
    # the arms ran what they are named after (launch counters of the library): batched reductions exist exactly in the deferred arm, stand-alone
    # column-sum passes exactly in the unfused-bias arm
    assert (launches["colsum_stage2_batched"] > 0) == _autograd._DEFER_LN_REDUCE and (launches["colsum"] > 40) == (not _autograd._FUSED_BIAS_GRAD), launches
    assert (launches["gemm_bf16_tn_splitk_group"] >= 12) == (_autograd._GROUPED_WGRAD and _autograd._FUSED_BIAS_GRAD), launches
    print(json.dumps({"workload": "CLIP ViT-B/16 training step (fwd + contrastive loss + bwd + SGD), synthetic" + ("" if a.tower == "both" else f" -- {a.tower.upper()} TOWER ALONE (tflops / mfma_frac do not apply)"), "batch": a.batch,
                      "ms_per_step": round(ms, 3), "pairs_per_s": round(a.batch / ms * 1e3, 1), "gflop_per_pair": gf,
                      "tflops": round(a.batch * gf / ms, 1) if a.tower == "both" else None,
                      "mfma_frac": round(a.batch * gf / ms / 2500.0, 4) if a.tower == "both" else None,  # (one tower alone: the whole-pair FLOP count does not apply)
                      "fsdp": ({"units": fsdp_units, "use_orig_params": a.fsdp_orig_params, "sharding": str(wrapped.sharding_strategy), "ranks": 1} if wrapped is not None else None),
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "fused_bias_grad": _autograd._FUSED_BIAS_GRAD, "deferred_ln_reduce": _autograd._DEFER_LN_REDUCE, "dh_dtype": "bf16" if _autograd._BF16_DH else "f32", "grouped_wgrad": _autograd._GROUPED_WGRAD, "launches_per_step": launches, "losses": [round(x, 4) for x in losses]}))

    if a.fsdp:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
