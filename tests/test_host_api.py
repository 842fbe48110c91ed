"""CPU-only checks of the host side: C-ABI symbols, drop-in API surface (signatures, state_dict keys, errors),
no-CPU-fallback behaviour."""
import inspect
import re
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def built_lib():
    from multimodal_amd import build

    return build.build()  # hipcc cross-compiles gfx950 without a GPU


def test_capi_exports_every_declared_symbol(built_lib):
    """libmmamd.so loads and exports every prototype of include/mmamd.h (the drop-in surface) and include/mmamd_debug.h (bench / experiment
    hooks, split off in r05); the ctypes table lists them all, and no debug hook hides in the product header."""
    import ctypes

    from multimodal_amd import _lib

    header = (ROOT / "include" / "mmamd.h").read_text()
    debug = (ROOT / "include" / "mmamd_debug.h").read_text()
    product = set(re.findall(r"\b(mmamd_[a-z0-9_]+)\s*\(", header))
    hooks = set(re.findall(r"\b(mmamd_[a-z0-9_]+)\s*\(", debug))
    assert product and hooks and not (product & hooks)
    assert not any(n.startswith("mmamd_debug_") or n.startswith("mmamd_timer_") or n in ("mmamd_set_gemm_variant", "mmamd_stream_set_cus") for n in product)
    declared = product | hooks
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    handle = ctypes.CDLL(str(built_lib))
    for name in declared:
        assert hasattr(handle, name), name
    assert _lib.lib().mmamd_abi_version() == _lib.ABI_VERSION


def test_argument_errors_do_not_need_a_gpu(built_lib):
    """Argument validation happens before any launch: callable on a CPU-only box."""
    from multimodal_amd import _lib

    L = _lib.lib()
    assert L.mmamd_gemm_bf16(None, 0, None, 0, None, None, 0, None, 0, 1, 4, 8, 64, 0, None) == -1
    assert b"null" in L.mmamd_last_error()
    assert L.mmamd_attention_fwd(None, None, 1, 300, 1, 0, 0.125, None) < 0
    assert L.mmamd_layernorm(None, 0, None, None, None, 1, 1, 64, 1e-5, None) < 0


def test_no_cpu_fallback():
    from multimodal_amd import ops
    from multimodal_amd.models.clip import CLIPTextEncoder, CLIPViTEncoder
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature

    vit = CLIPViTEncoder(embedding_dim=8, heads=1, layers=1, patch_size=16, image_size=32, width=64).eval()
    with pytest.raises(ops.MmamdError):
        vit(torch.zeros(1, 3, 32, 32))
    txt = CLIPTextEncoder(embedding_dim=8, context_length=8, vocab_size=32, width=64, heads=1, layers=1).eval()
    with pytest.raises(ops.MmamdError):
        txt(torch.zeros(1, 8, dtype=torch.long))
    with pytest.raises(ops.MmamdError):
        ContrastiveLossWithTemperature()(torch.zeros(2, 4), torch.zeros(2, 4))


def test_training_forward_needs_the_gpu_and_other_families_still_refuse():
    """CLIP, FLAVA (encoders, pre-training heads, losses) and CoCa have a differentiable (training) forward on the HIP kernels; like
    everything else it has no CPU path -- post-norm encoder AND decoder layers (the reference's default; trainable since r05) included."""
    from multimodal_amd import ops
    from multimodal_amd.models.clip import CLIPViTEncoder
    from multimodal_amd.models.flava.transformer import TransformerEncoder as FlavaEncoder
    from multimodal_amd.modules.layers.transformer import TransformerEncoder as LayersEncoder
    from multimodal_amd.modules.losses.flava import FLAVAPretrainingLoss

    vit = CLIPViTEncoder(embedding_dim=8, heads=1, layers=1, patch_size=16, image_size=32, width=64).train()
    with pytest.raises(ops.MmamdError, match="no CPU"):
        vit(torch.zeros(1, 3, 32, 32))
    with pytest.raises(ops.MmamdError, match="no CPU"):
        FlavaEncoder(1, 128, 2, 256, activation=torch.nn.GELU, norm_first=True).train()(torch.zeros(1, 4, 128))
    with pytest.raises(ops.MmamdError, match="no CPU"):  # post-norm (norm_first=False, the reference's default)
        FlavaEncoder(1, 128, 2, 256, activation=torch.nn.GELU).train()(torch.zeros(1, 4, 128))
    with pytest.raises(ops.MmamdError, match="no CPU"):  # CoCa's layers train too
        LayersEncoder(1, 128, 2, 256, activation=torch.nn.GELU, norm_first=True).train()(torch.zeros(1, 4, 128))
    with pytest.raises(ops.MmamdError, match="no CPU"):
        LayersEncoder(1, 128, 2, 256, activation=torch.nn.GELU).train()(torch.zeros(1, 4, 128))
    from multimodal_amd.modules.layers.transformer import TransformerDecoder

    with pytest.raises(ops.MmamdError, match="no CPU"):  # post-norm decoder layers
        TransformerDecoder(1, 128, 2, 256, activation=torch.nn.GELU, use_cross_attention=False).train()(torch.zeros(1, 4, 128))
    # a stand-alone layer in training is a differentiable call now (it used to raise NotImplementedError): no CPU path either
    from multimodal_amd.modules.layers.transformer import TransformerEncoderLayer

    with pytest.raises(ops.MmamdError, match="no CPU"):
        TransformerEncoderLayer(128, 2, 256, activation=torch.nn.GELU, norm_first=True).train()(torch.zeros(1, 4, 128))
    with pytest.raises(ops.MmamdError, match="no CPU"):  # the pre-training heads are differentiable too: still no CPU path
        FLAVAPretrainingLoss(hidden_size=128, text_vocab_size=64, image_vocab_size=64)(
            image_masked_sequence=torch.zeros(1, 5, 128, requires_grad=True), mim_labels=torch.zeros(1, 4, dtype=torch.long))


def test_input_guards_match_reference():
    """ValueErrors of image_encoder.py:83-88, text_encoder.py:114-117, contrastive_loss…:172-175."""
    from multimodal_amd.models.clip import CLIPTextEncoder, CLIPViTEncoder
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature

    vit = CLIPViTEncoder(embedding_dim=4, heads=2, layers=1, patch_size=2, image_size=16, width=2).eval()
    with pytest.raises(ValueError):
        vit(torch.ones(2, 3, 5, 5))
    with pytest.raises(ValueError):
        vit(torch.ones(2, 2, 16, 16))
    txt = CLIPTextEncoder(embedding_dim=4, heads=2, width=64, layers=1).eval()
    with pytest.raises(ValueError):
        txt(torch.ones(2, 78, dtype=torch.long))
    with pytest.raises(ValueError):
        ContrastiveLossWithTemperature(logit_scale_max=None, logit_scale_min=None)
    with pytest.raises(ValueError):  # reference quirk kept: the default min ln(1) = 0.0 is falsy
        ContrastiveLossWithTemperature(logit_scale_max=None)
    loss = ContrastiveLossWithTemperature()
    assert loss.logit_scale.shape == () and abs(float(loss.logit_scale) - np.log(1 / 0.07)) < 1e-6
    p = torch.nn.Parameter(torch.tensor(1.5))
    assert ContrastiveLossWithTemperature(logit_scale=p).logit_scale is p


def test_signatures_match_reference_surface():
    from multimodal_amd.models.clip import CLIP, CLIPTextEncoder, CLIPViTEncoder, clip_vit_b16, clip_vit_b32, clip_vit_l14
    from multimodal_amd.modules.losses import contrastive_loss_with_temperature as L
    from multimodal_amd.utils.distributed import BackpropType, gather_tensor

    assert list(inspect.signature(CLIPViTEncoder.__init__).parameters)[1:] == ["embedding_dim", "patch_size", "image_size", "width", "heads", "layers"]
    sig = inspect.signature(CLIPTextEncoder.__init__).parameters
    assert [(k, v.default) for k, v in list(sig.items())[1:]] == [("embedding_dim", 512), ("context_length", 77), ("vocab_size", 49408),
                                                                  ("width", 512), ("dim_feedforward", 2048), ("heads", 8), ("layers", 12), ("use_clip_init", True)]
    assert list(inspect.signature(CLIPTextEncoder.forward).parameters) == ["self", "text", "return_hidden_state"]
    assert list(inspect.signature(CLIP.forward).parameters) == ["self", "features_a", "features_b"]
    assert list(inspect.signature(L.ContrastiveLossWithTemperature.forward).parameters) == ["self", "embeddings_a", "embeddings_b", "backprop_type", "cross_entropy_kwargs", "mask"]
    assert list(inspect.signature(L.contrastive_loss_with_temperature).parameters) == ["embeddings_a", "embeddings_b", "logit_scale", "mask", "backprop_type", "cross_entropy_kwargs"]
    assert [b.name for b in BackpropType] == ["GLOBAL", "LOCAL", "NONE"] and [b.value for b in BackpropType] == [0, 1, 2]
    assert list(inspect.signature(gather_tensor).parameters) == ["tensor", "backprop_type"]
    for f in (clip_vit_b16, clip_vit_b32, clip_vit_l14):
        assert list(inspect.signature(f).parameters) == ["pretrained"]


def test_state_dict_keys_and_shapes_match_reference(golden):
    """Key names/order/shapes of the full-size models == what the reference produced (fixture key lists; 301 tensors for B/16)."""
    from multimodal_amd.models.clip import clip_vit_b16

    z = golden("clip_b16_b4.npz")
    sd = clip_vit_b16().state_dict()
    assert list(sd.keys()) == [str(k) for k in z["keys"]] and len(sd) == 301
    assert sd["encoder_a.conv.weight"].shape == (768, 3, 16, 16)
    assert sd["encoder_a.encoder.layers.11.self_attn.in_proj_weight"].shape == (2304, 768)
    assert sd["encoder_a.projection"].shape == (768, 512)
    assert sd["encoder_b.token_embedding.weight"].shape == (49408, 512)
    assert sd["encoder_b.projection.weight"].shape == (512, 512)
    assert not list(clip_vit_b16().buffers())


def test_vit_layers_share_initial_weights_like_torch_transformer_encoder():
    from multimodal_amd.models.clip import CLIPViTEncoder

    v = CLIPViTEncoder(embedding_dim=8, heads=1, layers=3, patch_size=16, image_size=32, width=64)
    w = [l.linear1.weight for l in v.encoder.layers]
    assert torch.equal(w[0], w[1]) and torch.equal(w[0], w[2]) and w[0].data_ptr() != w[1].data_ptr()


def test_text_clip_init_std():
    """reference tests/models/clip/test_text_encoder.py:42-94"""
    from multimodal_amd.models.clip import CLIPTextEncoder

    torch.manual_seed(1234)
    t = CLIPTextEncoder(embedding_dim=50, heads=2)
    assert abs(torch.std(t.token_embedding.weight).item() - 0.02) < 1e-4
    assert abs(torch.std(t.positional_embedding).item() - 0.01) < 1e-3
    for layer in t.encoder.layers:
        assert abs(torch.std(layer.self_attn.in_proj_weight).item() - 0.0442) < 5e-3
        assert abs(torch.std(layer.self_attn.out_proj.weight).item() - 0.0090) < 5e-3
        assert abs(torch.std(layer.linear1.weight).item() - 0.0313) < 5e-3
        assert abs(torch.std(layer.linear2.weight).item() - 0.0090) < 5e-3
    m = CLIPTextEncoder(context_length=4, heads=2, width=64, layers=1).build_attention_mask()
    inf = float("inf")
    assert torch.equal(m, torch.tensor([[0, -inf, -inf, -inf], [0, 0, -inf, -inf], [0, 0, 0, -inf], [0, 0, 0, 0.0]]))


def test_load_module_from_url_reads_a_reference_layout_checkpoint(golden, tmp_path):
    """utils/common.py:99-107 without iopath: a checkpoint file in the reference's state_dict layout (here the midsize fixture's
    weights, written with torch.save like the published .pt files) loads with strict=True; a wrong layout raises like torch does."""
    from multimodal_amd.models.clip import CLIP, CLIPTextEncoder, CLIPViTEncoder
    from multimodal_amd.utils.common import load_module_from_url
    from tests._util import fixture_sd

    z = golden("midsize.npz")
    sd = {k: torch.from_numpy(v) for k, v in fixture_sd(z).items()}
    path = tmp_path / "clip_midsize.pt"
    torch.save(sd, path)
    vit = CLIPViTEncoder(embedding_dim=64, heads=2, layers=2, patch_size=16, image_size=64, width=128)
    txt = CLIPTextEncoder(embedding_dim=64, context_length=77, vocab_size=1000, width=128, dim_feedforward=256, heads=2, layers=2)
    clip = CLIP(vit, txt)
    load_module_from_url(clip, str(path))
    for k, v in clip.state_dict().items():
        assert torch.equal(v, sd[k]), k
    sd.pop("encoder_a.projection")
    torch.save(sd, path)
    with pytest.raises(RuntimeError, match="Missing key"):
        load_module_from_url(clip, str(path))
    load_module_from_url(clip, str(path), strict=False)
