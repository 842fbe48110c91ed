"""FLAVA's text tower factory: same name, keyword contract and defaults as the reference's `flava_text_encoder`
(torchmultimodal/models/flava/text_encoder.py:21-71), so configs and call sites written for it keep working."""
from functools import partial
from typing import Callable

from torch import nn

from ...modules.encoders.bert_text_encoder import BERTTextEncoder
from ...modules.layers.normalizations import Fp32LayerNorm
from ...modules.layers.text_embedding import BERTTextEmbeddings
from ...modules.losses.flava import Pooler
from .transformer import init_transformer_weights, TransformerEncoder


def flava_text_encoder(num_hidden_layers: int = 12, hidden_size: int = 768, num_attention_heads: int = 12, intermediate_size: int = 3072,
                       intermediate_activation: Callable[..., nn.Module] = nn.GELU, layer_norm_eps: float = 1e-12, dropout: float = 0.0,
                       vocab_size: int = 30522, pad_token_id: int = 0, type_vocab_size: int = 2, max_position_embeddings: int = 512,
                       initializer_range: float = 0.02) -> BERTTextEncoder:
    """BERT-base shaped by default: 12 pre-norm layers of width 768 behind word + position + token-type embeddings, an fp32 LayerNorm on the
    way out and a tanh pooler over the CLS row.

    The four parts are built in the order embeddings -> encoder -> final norm -> pooler: every nn.Embedding / nn.Linear draws its initial
    values from torch's global generator when it is constructed, and the seeded-construction checks (tests/test_host_api_flava.py: all 511
    tensors of flava_model() at seed 0 equal the reference's) depend on consuming the generator in the reference's order before
    BERTTextEncoder re-initialises the lot with N(0, initializer_range)."""
    shared = {"layer_norm_eps": layer_norm_eps, "dropout": dropout}  # the one epsilon / dropout rate every part of the tower uses
    parts = {
        "embeddings": BERTTextEmbeddings(hidden_size, vocab_size, pad_token_id, max_position_embeddings, type_vocab_size, **shared),
        "encoder": TransformerEncoder(num_hidden_layers, hidden_size, num_attention_heads, intermediate_size, activation=intermediate_activation,
                                      norm_first=True, **shared),
        "layernorm": Fp32LayerNorm(hidden_size, eps=layer_norm_eps),
        "pooler": Pooler(hidden_size),
    }
    return BERTTextEncoder(weight_init_fn=partial(init_transformer_weights, initializer_range=initializer_range), **parts)
