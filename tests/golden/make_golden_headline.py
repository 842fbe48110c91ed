"""Headline-size parity fixtures from the REFERENCE itself (mechanism: make_golden.py).  Build container only:

    python -m tests.golden.make_golden_headline [b16] [l14] [coca] [flava]

  clip_b16_b256.npz   cfg 2 AT FULL SIZE: reference clip_vit_b16 (seed-0 default init), the SURVEY 8d batch of 256 pairs, CPU fp32:
                      emb_a / emb_b [256,512], both [256,256] logit blocks, the loss; and the reference's OWN bf16-CPU run
                      (model.to(bfloat16), bf16 images): its logits and its argmax agreement with its fp32 run — the rate the HIP path
                      is compared with ("report the unfiltered agreement rate next to the reference's own bf16-CPU rate")
  clip_l14_b32.npz    cfg 3 model (clip_vit_l14) at B = 32: same fields
  flava_full_b16.npz  cfg 4 model (flava_model(), 241 M parameters) at B = 16: FLAVAModel.forward with a patch mask and masked / padded text
                      (projected embeddings, CLS rows of all five encoders' outputs, multimodal pooler) + the global contrastive loss
  coca_l14_b8.npz     cfg 5 model (coca_vit with the coca_vit_l_14 arguments and cascaded_pooler=False: the parallel pooler, SURVEY 8a
                      note) at B = 8 with padded captions: pooled embeddings, both losses of CoCaForPretraining, and of the
                      [8,76,49408] vocabulary logits (120 MB) a strided column sample, the per-row argmax / max / logsumexp and the
                      logit of the label token
Weights are not stored: the tests re-create them from the seed through the drop-in modules and verify the per-tensor checksums.
Also writes profiles/r02_reference_cpu.json: the wall time of the reference forward + loss on this container's CPU.
"""
from __future__ import annotations

import json
import math
import os
import sys
import time
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.golden import _ref_shim  # noqa: E402
from tests.golden.make_golden import checksums, seed  # noqa: E402

OUT = Path(__file__).resolve().parent
warnings.filterwarnings("ignore")

COCA_L14 = dict(vision_patch_size=14, vision_n_layer=24, vision_n_head=16, vision_dim_feedforward=4096, vision_include_cls_embed=False,
                vocab_size=49408, num_text_positions=77, text_hidden_dim=768, text_n_layer=12, text_n_head=12, text_dim_feedforward=3072,
                text_output_dim=768, fusion_n_layer=12, fusion_n_head=12, fusion_dim_feedforward=3072,
                multimodal_output_projection_dim=49408, pooler_input_embed_dim=1024, pooler_output_embed_dim=768, pooler_n_head=8)


def tnp(t):
    return t.detach().float().numpy().copy()


def agreement(x, y):
    return float((x.argmax(1) == y.argmax(1)).mean())


def clip_case(name, ref_factory, my_factory, B, timing):
    from torchmultimodal.modules.losses.contrastive_loss_with_temperature import contrastive_loss_with_temperature as ref_loss_fn

    from multimodal_amd.utils.synthetic import clip_batch

    seed(0)
    ref = ref_factory().eval()
    seed(0)
    mine = my_factory()
    rs, ms = ref.state_dict(), mine.state_dict()
    assert list(rs.keys()) == list(ms.keys()) and all(torch.equal(rs[k], ms[k]) for k in rs), name
    del mine, ms
    images, ids = clip_batch(B)
    scale = torch.nn.Parameter(torch.tensor(math.log(1 / 0.07), dtype=torch.float32))
    times = []
    with torch.no_grad():
        for it in range(3):
            t0 = time.perf_counter()
            out = ref(images, ids)
            lo = ref_loss_fn(out.embeddings_a, out.embeddings_b, scale)
            times.append(time.perf_counter() - t0)
            print(name, "fp32 pass", it, f"{times[-1]:.2f} s", flush=True)
    k, s, a = checksums(ref)
    st = dict(emb_a=tnp(out.embeddings_a), emb_b=tnp(out.embeddings_b), loss=tnp(lo.loss), logits_a=tnp(lo.logits_a),
              logits_b=tnp(lo.logits_b), keys=k, sums=s, asums=a, images_sum=float(images.double().sum()), ids_sum=int(ids.sum()))
    # the reference's own reduced-precision CPU path: weights and images in bf16 (SURVEY 6 / BASELINE.md 4)
    ref16 = ref.to(torch.bfloat16)
    with torch.no_grad():
        t0 = time.perf_counter()
        o16 = ref16(images.to(torch.bfloat16), ids)
        l16 = ref_loss_fn(o16.embeddings_a.float(), o16.embeddings_b.float(), scale)
        t16 = time.perf_counter() - t0
    st.update(bf16_logits_a=tnp(l16.logits_a), bf16_logits_b=tnp(l16.logits_b), bf16_loss=tnp(l16.loss),
              bf16_agree_a=agreement(tnp(l16.logits_a), st["logits_a"]),
              bf16_agree_b=agreement(tnp(l16.logits_b), st["logits_b"]),
              bf16_max_logit_diff=float(max(np.abs(tnp(l16.logits_a) - st["logits_a"]).max(), np.abs(tnp(l16.logits_b) - st["logits_b"]).max())))
    np.savez_compressed(OUT / f"{name}.npz", **st)
    med = sorted(times[1:])[len(times[1:]) // 2] if len(times) > 1 else times[0]
    timing[name] = {"model": ref_factory.__name__, "batch": B, "dtype": "fp32", "threads": torch.get_num_threads(),
                    "cores": os.cpu_count(), "step_s": [round(t, 3) for t in times], "step_s_median_after_warmup": round(med, 3),
                    "pairs_per_s": round(B / med, 3), "bf16_cpu_step_s": round(t16, 3), "loss": float(lo.loss),
                    "bf16_argmax_agreement": [st["bf16_agree_a"], st["bf16_agree_b"]], "bf16_max_logit_diff": st["bf16_max_logit_diff"]}
    print(name, timing[name], flush=True)


def coca_case(timing):
    from torchmultimodal.models.coca.coca_model import coca_vit, CoCaForPretraining

    from multimodal_amd.models.coca.coca_model import coca_vit as my_coca_vit

    B = 8
    seed(0)
    model = coca_vit(**COCA_L14, cascaded_pooler=False).eval()
    seed(0)
    mine = my_coca_vit(**COCA_L14, cascaded_pooler=False)
    rs, ms = model.state_dict(), mine.state_dict()
    assert list(rs.keys()) == list(ms.keys()) and all(torch.equal(rs[k], ms[k]) for k in rs), "coca l14 init"
    del mine, ms
    g = torch.Generator().manual_seed(4321)
    images = torch.randn(B, 3, 224, 224, generator=g)
    texts = torch.randint(1, 49407, (B, 77), generator=g)
    lens = [77, 60, 41, 77, 23, 9, 77, 52]
    for i, n in enumerate(lens):
        texts[i, n:] = 0  # pad_idx 0
    pre = CoCaForPretraining(model).eval()
    with torch.no_grad():
        t0 = time.perf_counter()
        out = model(images, texts)
        t_model = time.perf_counter() - t0
        losses = pre(images, texts)
    mm = out.multimodal_embeddings  # [B, 76, 49408]
    labels = texts[:, 1:]
    lse = torch.logsumexp(mm, dim=-1)
    lab_logit = torch.gather(mm, 2, labels.unsqueeze(-1)).squeeze(-1)
    top2 = torch.topk(mm, 2, dim=-1).values
    k, s, a = checksums(model)
    st = dict(texts=texts.numpy(), images_sum=float(images.double().sum()), image_pooled_output=tnp(out.image_pooled_output),
              text_pooled_output=tnp(out.text_pooled_output), mm_cols=np.arange(0, 49408, 64), mm_sample=tnp(mm[:, :, ::64]),
              mm_argmax=mm.argmax(-1).numpy(), mm_max=tnp(top2[..., 0]), mm_second=tnp(top2[..., 1]), mm_lse=tnp(lse),
              mm_label_logit=tnp(lab_logit), loss_contrastive=tnp(losses["contrastive"]), loss_captioning=tnp(losses["captioning"]),
              logit_scale=tnp(pre.contrastive_loss.logit_scale), keys=k, sums=s, asums=a)
    np.savez_compressed(OUT / "coca_l14_b8.npz", **st)
    timing["coca_l14_b8"] = {"model": "coca_vit(l14 arguments, cascaded_pooler=False)", "batch": B, "dtype": "fp32",
                             "threads": torch.get_num_threads(), "cores": os.cpu_count(), "model_forward_s": round(t_model, 3),
                             "samples_per_s": round(B / t_model, 3), "loss_contrastive": float(losses["contrastive"]),
                             "loss_captioning": float(losses["captioning"])}
    print("coca_l14_b8", timing["coca_l14_b8"], flush=True)


def flava_case(timing):
    """cfg 4 model at B = 16: flava_model() (seed 0), patch mask + masked / padded text, the whole FLAVAModel.forward + ITC loss."""
    from torchmultimodal.models.flava.model import flava_model
    from torchmultimodal.modules.losses.flava import FLAVAGlobalContrastiveLoss

    from multimodal_amd.models.flava.model import flava_model as my_flava_model

    B = 16
    seed(0)
    model = flava_model().eval()
    seed(0)
    mine = my_flava_model()
    rs, ms = model.state_dict(), mine.state_dict()
    assert list(rs.keys()) == list(ms.keys()) and all(torch.equal(rs[k], ms[k]) for k in rs), "flava init"
    del mine, ms
    g = torch.Generator().manual_seed(2024)
    image = torch.randn(B, 3, 224, 224, generator=g)
    text = torch.randint(1, 30500, (B, 77), generator=g)
    for i, n in enumerate([77, 60, 41, 77, 23, 9, 77, 52] * 2):
        text[i, n:] = 0
    text_masked = text.clone()
    text_masked[torch.rand(B, 77, generator=g) < 0.15] = 103
    text_masked[text == 0] = 0
    patches_mask = torch.randint(0, 2, (B, 196), generator=g)
    with torch.no_grad():
        t0 = time.perf_counter()
        out = model(image, text, image_patches_mask=patches_mask, text_masked=text_masked, skip_unmasked_mm_encoder=True)
        t_fwd = time.perf_counter() - t0
        lo = FLAVAGlobalContrastiveLoss().eval()(out.projected_image_embeddings, out.projected_text_embeddings, torch.ones(B, dtype=torch.bool))
    k, s, a = checksums(model)
    st = dict(text=text.numpy(), text_masked=text_masked.numpy(), patches_mask=patches_mask.numpy(), image_sum=float(image.double().sum()),
              proj_image=tnp(out.projected_image_embeddings), proj_text=tnp(out.projected_text_embeddings),
              image_cls=tnp(out.image.last_hidden_state[:, 0]), text_cls=tnp(out.text.last_hidden_state[:, 0]),
              image_masked_cls=tnp(out.image_masked.last_hidden_state[:, 0]), text_masked_cls=tnp(out.text_masked.last_hidden_state[:, 0]),
              mm_masked_cls=tnp(out.multimodal_masked.last_hidden_state[:, 0]), mm_masked_pooler=tnp(out.multimodal_masked.pooler_output),
              mm_masked_hidden_mean=float(out.multimodal_masked.last_hidden_state.double().mean()),
              itc_loss=tnp(lo.loss), itc_image_logits=tnp(lo.image_logits), itc_text_logits=tnp(lo.text_logits), keys=k, sums=s, asums=a)
    np.savez_compressed(OUT / "flava_full_b16.npz", **st)
    timing["flava_full_b16"] = {"model": "flava_model()", "batch": B, "dtype": "fp32", "threads": torch.get_num_threads(), "cores": os.cpu_count(),
                                "model_forward_s": round(t_fwd, 3), "samples_per_s": round(B / t_fwd, 3), "itc_loss": float(lo.loss)}
    print("flava_full_b16", timing["flava_full_b16"], flush=True)


def main():
    assert _ref_shim.reference_available(), "needs /root/reference"
    _ref_shim.install()
    from torchmultimodal.models.clip.model import clip_vit_b16 as ref_b16, clip_vit_l14 as ref_l14

    from multimodal_amd.models.clip import clip_vit_b16, clip_vit_l14

    torch.set_num_threads(8)
    which = set(sys.argv[1:]) or {"b16", "l14", "coca", "flava"}
    path = ROOT / "profiles" / "r02_reference_cpu.json"
    timing = json.loads(path.read_text()) if path.exists() else {}
    timing["_host"] = {"cpu": "build container", "os_cpu_count": os.cpu_count(), "torch_threads": 8, "torch": torch.__version__,
                       "what": "the reference itself (imported from /root/reference under tests/golden/_ref_shim.py), eval, no_grad, "
                               "fp32, forward of both towers + contrastive_loss_with_temperature on the SURVEY 8d synthetic batch"}
    if "b16" in which:
        clip_case("clip_b16_b256", ref_b16, clip_vit_b16, 256, timing)
    if "l14" in which:
        clip_case("clip_l14_b32", ref_l14, clip_vit_l14, 32, timing)
    if "coca" in which:
        coca_case(timing)
    if "flava" in which:
        flava_case(timing)
    path.write_text(json.dumps(timing, indent=1) + "\n")
    print("written", path)


if __name__ == "__main__":
    main()
