"""CPU-side checks of the FLAVA drop-in modules: constructor / state_dict / seeded-init parity with the reference (through
the committed fixtures), output records, and loud failures where the MI355X path has no implementation (no CPU fallback)."""
import numpy as np
import pytest
import torch
from torch import nn

from multimodal_amd import ops
from tests._util import assert_checksums, fixture_sd
from tests.conftest import set_rng_seed

SMALL_KW = dict(image_hidden_size=128, image_num_attention_heads=2, image_num_hidden_layers=2, image_intermediate_size=256,
                image_size=32, patch_size=16, text_hidden_size=128, text_num_attention_heads=2, text_num_hidden_layers=2,
                text_intermediate_size=256, vocab_size=200, max_position_embeddings=32, multimodal_hidden_size=128,
                multimodal_num_attention_heads=2, multimodal_num_hidden_layers=2, multimodal_intermediate_size=256,
                text_and_image_proj_size=64)


def test_flava_model_state_dict_and_seeded_init_match_reference(golden):
    from multimodal_amd.models.flava.model import flava_model

    z = golden("flava_full_b2.npz")
    set_rng_seed(0)
    model = flava_model()  # 241 M parameters: keys, order, and every seeded initial tensor equal the reference's
    assert_checksums(model, z)
    zs = golden("flava_small.npz")
    small = flava_model(**SMALL_KW)
    missing = small.load_state_dict({k: torch.from_numpy(v) for k, v in fixture_sd(zs).items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys


def test_flava_layer_state_dict_matches_reference_kat(golden):
    from multimodal_amd.models.flava.transformer import TransformerEncoderLayer

    z = golden("flava_layer_kat.npz")
    set_rng_seed(4)
    x = torch.randn(1, 2, 2, 2, 2)
    layer = TransformerEncoderLayer(2, 1, 2, norm_first=True)
    ref = fixture_sd(z)
    assert list(layer.state_dict().keys()) == list(ref.keys())
    assert np.array_equal(x.numpy(), z["x"])
    for k, v in layer.state_dict().items():  # same RNG consumption order as the reference layer
        assert np.array_equal(v.numpy(), ref[k]), k


def test_output_records_and_factories():
    from multimodal_amd.models.flava.image_encoder import flava_image_encoder, ImageTransformer
    from multimodal_amd.models.flava.model import FLAVAModel, FLAVAOutput, flava_model, flava_multimodal_encoder
    from multimodal_amd.models.flava.text_encoder import flava_text_encoder
    from multimodal_amd.models.flava.transformer import FLAVATransformerWithoutEmbeddings
    from multimodal_amd.modules.encoders.bert_text_encoder import BERTTextEncoder
    from multimodal_amd.modules.layers.transformer import TransformerOutput
    from multimodal_amd.modules.losses.flava import FLAVAGlobalContrastiveLoss, FLAVAGlobalContrastiveLossOutput, Pooler
    import dataclasses

    assert TransformerOutput._fields == ("last_hidden_state", "pooler_output", "hidden_states", "attentions", "image_labels",
                                         "current_key_values")
    assert FLAVAOutput._fields == ("image", "image_masked", "text", "text_masked", "multimodal", "multimodal_masked",
                                   "projected_image_embeddings", "projected_text_embeddings")
    assert FLAVAOutput() == FLAVAOutput(*([None] * 8))
    assert [f.name for f in dataclasses.fields(FLAVAGlobalContrastiveLossOutput)] == [
        "text_embedding", "image_embedding", "logit_scale", "image_logits", "text_logits", "image_loss", "text_loss", "loss"]
    m = flava_model(**SMALL_KW)
    assert isinstance(m, FLAVAModel) and isinstance(m.image_encoder, ImageTransformer)
    assert isinstance(m.text_encoder, BERTTextEncoder) and isinstance(m.mm_encoder, FLAVATransformerWithoutEmbeddings)
    assert isinstance(m.image_encoder.pooler, Pooler) and m.image_encoder.embeddings.mask_token is not None
    assert flava_image_encoder(hidden_size=128, num_attention_heads=2, num_hidden_layers=1, intermediate_size=256, image_size=32).embeddings.mask_token is None
    assert flava_text_encoder(num_hidden_layers=1, hidden_size=128, num_attention_heads=2, intermediate_size=256, vocab_size=50).embeddings.pad_token_id == 0
    assert flava_multimodal_encoder(hidden_size=128, num_attention_heads=2, num_hidden_layers=1, intermediate_size=256).cls_token.shape == (1, 1, 128)
    loss = FLAVAGlobalContrastiveLoss()
    assert abs(float(loss.logit_scale) - np.log(1 / 0.07)) < 1e-6
    p = nn.Parameter(torch.tensor(1.5))
    assert FLAVAGlobalContrastiveLoss(logit_scale=p).logit_scale is p


@torch.no_grad()  # inference contract of the eval-mode modules (the train-mode calls below enable grad explicitly)
def test_flava_fails_loudly_without_a_gpu_or_an_implementation():
    from multimodal_amd.models.flava.image_encoder import PatchEmbeddings
    from multimodal_amd.models.flava.model import flava_model
    from multimodal_amd.models.flava.transformer import TransformerEncoder
    from multimodal_amd.modules.layers.attention import MultiHeadAttention
    from multimodal_amd.modules.layers.mlp import MLP
    from multimodal_amd.modules.losses.flava import FLAVAGlobalContrastiveLoss

    m = flava_model(**SMALL_KW).eval()
    with pytest.raises(ops.MmamdError, match="no CPU"), torch.no_grad():
        m(torch.randn(1, 3, 32, 32), torch.randint(1, 200, (1, 16)))
    # eval mode + grad mode on: inference like the reference's own tests run it -> reaches the device check, not a refusal
    from multimodal_amd import _autograd

    _autograd._warned_detached.clear()
    with pytest.raises(ops.MmamdError, match="no CPU"), torch.enable_grad(), pytest.warns(UserWarning, match="NOT attached"):
        m(torch.randn(1, 3, 32, 32), torch.randint(1, 200, (1, 16)))
    with pytest.raises(ValueError, match="doesn't match model"):
        m.image_encoder(torch.randn(1, 3, 48, 48))
    with pytest.raises(ValueError, match="pixel_values"):
        m.image_encoder(None)
    with pytest.raises(ValueError, match="input_ids or inputs_embeds"):
        m.text_encoder()
    with pytest.raises(ValueError, match="hidden_states"):
        m.mm_encoder(None)
    with pytest.raises(ops.MmamdError, match="no CPU"), torch.enable_grad():  # FLAVA trains on the HIP kernels; there is still no CPU path
        m.train()(torch.randn(1, 3, 32, 32), torch.randint(1, 200, (1, 16)))
    m.eval()
    with pytest.raises(ops.MmamdError):
        FLAVAGlobalContrastiveLoss()(torch.randn(2, 8), torch.randn(2, 8), torch.ones(2, dtype=torch.bool))
    with pytest.raises(ops.MmamdError, match="no CPU|HIP device"):  # head_mask is served by the general attention kernel (tests/test_gpu_flava.py): no CPU path either
        TransformerEncoder(1, 128, 2, 256).eval()(torch.randn(1, 4, 128), head_mask=torch.ones(1))
    # nn.ReLU MLPs (classifier heads) plan onto the exact-fp32 row path; activations without any kernel still raise
    from multimodal_amd.modules.layers.mlp import ACT_RELU_EXACT

    assert [a for _, a in MLP(128, 128, 256, dropout=0.0).plan()] == [ACT_RELU_EXACT, ops.ACT_NONE]
    with pytest.raises(ops.MmamdError, match="activation Tanh"):
        MLP(128, 128, 256, dropout=0.0, activation=nn.Tanh).plan()
    with pytest.raises(ops.MmamdError, match="no CPU|HIP device"):
        MLP(128, 128, 256, dropout=0.0).eval()(torch.randn(2, 128))
    assert [a for _, a in MLP(128, 128, [256, 256], dropout=0.0, activation=nn.GELU).plan()] == [ops.ACT_GELU_ERF, ops.ACT_GELU_ERF, ops.ACT_NONE]
    with pytest.raises(ValueError, match="multiple of the number of attention heads"):
        MultiHeadAttention(130, 130, 4)
    assert MLP(8, 4, 16).model[0].weight.shape == (16, 8) and isinstance(MLP(8, 4, 16).model[2], nn.Dropout)
    assert PatchEmbeddings(32, 16, 3, 64).num_patches == 4


def test_pretraining_loss_state_dict_and_records(golden):
    import dataclasses

    from multimodal_amd.modules.losses.flava import (FLAVAPretrainingLoss, FLAVAPretrainingLossesCollection,
                                                     FLAVAPretrainingLossOutput, ITMLossOutput, MaskedPredictionHead,
                                                     MaskedPredictionLossOutput)

    z = golden("flava_pretrain_small.npz")
    loss = FLAVAPretrainingLoss(hidden_size=128, text_vocab_size=200, image_vocab_size=64)
    ref = fixture_sd(z)
    assert list(loss.state_dict().keys()) == list(ref.keys())
    loss.load_state_dict({k: torch.from_numpy(v) for k, v in ref.items()}, strict=True)
    assert [f.name for f in dataclasses.fields(FLAVAPretrainingLossesCollection)] == [
        "mmm_text_loss", "mmm_image_loss", "mim_loss", "mlm_loss", "itm_loss", "global_contrastive_loss"]
    assert [f.name for f in dataclasses.fields(FLAVAPretrainingLossOutput)][:7] == [
        "losses", "mlm_output", "mim_output", "mmm_text_output", "mmm_image_output", "itm_output", "global_contrastive_output"]
    assert FLAVAPretrainingLossOutput().losses.itm_loss is None
    assert [f.name for f in dataclasses.fields(ITMLossOutput)] == ["logits", "loss"] == [f.name for f in dataclasses.fields(MaskedPredictionLossOutput)]
    head = MaskedPredictionHead(hidden_size=16, vocab_size=10)
    assert head.decoder.bias is head.bias  # tied output bias, like the reference
    with pytest.raises(ops.MmamdError, match="transform_act_fn"):
        MaskedPredictionHead(hidden_size=128, vocab_size=8, transform_act_fn=torch.relu).run(torch.zeros(1, 128))
    with pytest.raises(AssertionError, match="itm labels"):
        loss.itm_loss.train()(torch.zeros(1, 2, 128), None)
    with pytest.raises(ops.MmamdError):
        loss.eval()(image_masked_sequence=torch.zeros(1, 5, 128), mim_labels=torch.zeros(1, 4, dtype=torch.long))  # CPU tensors
