"""CPU-side checks of the CoCa drop-in modules: state_dict / seeded-init parity with the reference, output records, loud failures."""
import numpy as np
import pytest
import torch
from torch import nn

from multimodal_amd import ops
from tests._util import assert_checksums
from tests.golden.make_golden import seed
from tests.golden.make_golden_coca import POOL96, randomize, SMALL


def test_coca_l14_state_dict_keys_and_shapes_match_reference(golden):
    from multimodal_amd.models.coca.coca_model import coca_vit_b_32, coca_vit_l_14

    z = golden("coca_l14_meta.npz")
    with torch.device("meta"):
        model = coca_vit_l_14()
        b32 = coca_vit_b_32()
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in z["keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in z["shapes"]]
    assert len(model.vision_pooler.poolers) == 2 and model.vision_pooler.poolers[0].query.shape == (256, 768)
    assert model.vision_pooler.poolers[0].attn.num_heads == 8  # 96-wide heads
    assert b32.text_decoder.text_projection.weight.shape == (512, 512) and "causal_mask" not in b32.state_dict()


@pytest.mark.parametrize("fixture,kw,cascaded,seed_v,prefix", [
    ("coca_small.npz", SMALL, False, 51, "par."), ("coca_small.npz", SMALL, True, 52, "cas."), ("coca_pool96.npz", POOL96, False, 53, "par.")])
def test_coca_seeded_init_matches_reference(golden, fixture, kw, cascaded, seed_v, prefix):
    from multimodal_amd.models.coca.coca_model import coca_vit

    z = golden(fixture)
    seed(seed_v)
    model = coca_vit(**kw, cascaded_pooler=cascaded)
    randomize(model, torch.Generator().manual_seed(seed_v + 1))
    assert_checksums(model, {"keys": z[prefix + "keys"], "sums": z[prefix + "sums"], "asums": z[prefix + "asums"]})


@torch.no_grad()  # inference contract of the eval-mode modules (the train-mode calls below enable grad explicitly)
def test_coca_records_and_loud_failures():
    from multimodal_amd.models.coca.coca_model import (coca_for_pretraining, coca_vit, CoCaForPretraining, CoCaModel,
                                                       CoCaModelWithHeads, MultimodalOutput)
    from multimodal_amd.models.coca.text_decoder import CoCaTextDecoder, CoCaTextEmbeddings
    from multimodal_amd.modules.encoders.vision_transformer import GlobalAveragePooler, vit_b_16
    from multimodal_amd.modules.layers.multi_head_attention import MHAWithCacheOutput, to_attn_mask
    from multimodal_amd.modules.layers.patch_embedding import PatchEmbeddings, PatchEmbeddingsOutput
    from multimodal_amd.modules.layers.transformer import TransformerEncoder
    from multimodal_amd.utils.attention import get_causal_attention_mask, get_extended_attention_mask

    assert MultimodalOutput._fields == ("image_pooled_output", "text_pooled_output", "multimodal_embeddings", "multimodal_pooled_embeddings")
    assert MHAWithCacheOutput._fields == ("attn_output", "past_key_value")
    assert PatchEmbeddingsOutput._fields == ("embeddings", "random_mask", "ids_restore")
    assert torch.equal(get_causal_attention_mask(3), torch.tril(torch.ones(3, 3)))
    assert get_extended_attention_mask(torch.ones(2, 5)).shape == (2, 1, 1, 5) and get_extended_attention_mask(torch.ones(2, 4, 5)).shape == (2, 1, 4, 5)
    with pytest.raises(ValueError):
        get_extended_attention_mask(torch.ones(5))
    pre = coca_for_pretraining(**SMALL, cascaded_pooler=False)
    assert isinstance(pre, CoCaForPretraining) and isinstance(pre.model, CoCaModel) and pre.caption_loss.ignore_index == 0
    m = coca_vit(**SMALL, cascaded_pooler=False).eval()
    with pytest.raises(ops.MmamdError, match="no CPU"), torch.no_grad():
        m(torch.randn(1, 3, 64, 64), torch.randint(1, 96, (1, 13)))
    from multimodal_amd import _autograd

    _autograd._warned_detached.clear()
    # eval mode + grad mode on: inference like the reference's own tests run it -> reaches the device check, not a refusal;
    # an INPUT that requires grad is refused (its gradient would be cut off silently)
    with pytest.raises(ops.MmamdError, match="no CPU"), torch.enable_grad(), pytest.warns(UserWarning, match="NOT attached"):
        m(torch.randn(1, 3, 64, 64), torch.randint(1, 96, (1, 13)))
    with pytest.raises(NotImplementedError, match="INPUT requires grad"), torch.enable_grad():
        m(torch.randn(1, 3, 64, 64, requires_grad=True), torch.randint(1, 96, (1, 13)))
    with pytest.raises(ValueError, match="doesn't match image size"):
        m.vision_encoder(torch.randn(1, 3, 32, 32))
    with pytest.raises(AssertionError):
        m.text_decoder(torch.randint(1, 96, (1, 9)))
    with pytest.raises(ops.MmamdError, match="no CPU"), torch.enable_grad():  # CoCa trains on the HIP kernels; there is still no CPU path
        m.train()(torch.randn(1, 3, 64, 64), torch.randint(1, 96, (1, 13)))
    m.eval()
    with pytest.raises(ValueError, match="divisible by patch size"):
        PatchEmbeddings(image_size=30, patch_size=16)
    # stochastic depth is built like the reference builds it (transformer.py:64-67,190-193): ONE StochasticDepth(mode="row") on both residual
    # branches of a layer, the rate growing linearly with depth; no parameters, so the state_dict is the dropout-free one
    from multimodal_amd.modules.layers.stochastic_depth import StochasticDepth

    sd_enc = TransformerEncoder(3, 128, 2, 256, drop_path_rate=0.2)
    assert [l.attention_dropout.p for l in sd_enc.layer] == pytest.approx([0.0, 0.1, 0.2])
    assert all(isinstance(l.attention_dropout, StochasticDepth) and l.attention_dropout is l.feedforward_dropout and l.attention_dropout.mode == "row"
               for l in sd_enc.layer)
    assert set(sd_enc.state_dict()) == set(TransformerEncoder(3, 128, 2, 256).state_dict())
    x_cpu = torch.randn(2, 3, 8)
    assert StochasticDepth(0.5, "row").eval()(x_cpu) is x_cpu and StochasticDepth(0.0, "row").train()(x_cpu) is x_cpu
    with pytest.raises(ops.MmamdError, match="boolean"):
        to_attn_mask(torch.zeros(2, 4, 4), False, 2, 4, 4)
    with pytest.raises(ops.MmamdError, match="per-head"):
        to_attn_mask(torch.ones(2, 3, 4, 4, dtype=torch.bool), False, 2, 4, 4)
    emb = CoCaTextEmbeddings(vocab_size=10, num_positions=5, embedding_dim=8)
    assert float(emb.cls_embedding[0]) == pytest.approx(0.01) and emb.token_embeddings.padding_idx == 0
    dec = CoCaTextDecoder(vocab_size=10, num_positions=5, embedding_dim=128, n_layer=1, n_head=2, dim_feedforward=256, output_dim=64)
    assert dec.causal_mask.dtype == torch.bool and dec.causal_mask.shape == (5, 5) and dec.text_projection.bias is None
    assert isinstance(vit_b_16().pooler, type(None)) and isinstance(GlobalAveragePooler(8, 4).head, nn.Linear)
    heads = CoCaModelWithHeads(m, nn.ModuleDict({"cls": nn.Linear(96, 3)}))
    assert list(heads.heads.keys()) == ["cls"]
