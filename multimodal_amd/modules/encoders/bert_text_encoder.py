"""Host-side mirror of torchmultimodal/modules/encoders/bert_text_encoder.py:17-123 (BERTTextEncoder): embeddings ->
transformer encoder -> optional LayerNorm -> optional pooler.  The padding mask (ids != pad) is built by a kernel as the
uint8 key mask the attention kernel consumes, instead of the reference's [B,1,1,S] float tensor."""
from __future__ import annotations

from typing import Callable, Optional

import torch
from torch import nn, Tensor

from ... import ops
from ..layers.transformer import TransformerOutput


class BERTTextEncoder(nn.Module):
    def __init__(
        self,
        embeddings: nn.Module,
        encoder: nn.Module,
        layernorm: Optional[nn.Module] = None,
        pooler: Optional[nn.Module] = None,
        weight_init_fn: Optional[Callable] = None,
    ) -> None:
        super().__init__()
        self.embeddings = embeddings
        self.encoder = encoder
        self.layernorm = layernorm
        self.pooler = pooler
        if weight_init_fn:
            self.apply(weight_init_fn)

    def forward(
        self,
        input_ids: Optional[Tensor] = None,
        attention_mask: Optional[Tensor] = None,
        token_type_ids: Optional[Tensor] = None,
        position_ids: Optional[Tensor] = None,
        inputs_embeds: Optional[Tensor] = None,
        return_attn_weights: bool = False,
        return_hidden_states: bool = False,
    ) -> TransformerOutput:
        if input_ids is None and inputs_embeds is None:
            raise ValueError("input_ids or inputs_embeds must not be None")
        ids = None
        if input_ids is not None:
            ids = input_ids if input_ids.is_contiguous() else input_ids.contiguous()
        if attention_mask is None:
            # only mask out padding tokens if no mask specified (reference :84-88); with inputs_embeds alone every position is attended (:86-87)
            if ids is not None and hasattr(self.embeddings, "pad_token_id"):
                key_mask = ops.key_mask(ids, pad_id=self.embeddings.pad_token_id)
            else:
                key_mask = None
        else:
            key_mask = ops.key_mask(attention_mask if attention_mask.is_contiguous() else attention_mask.contiguous())
        if key_mask is not None:
            key_mask._mmamd_key_mask = True  # already in kernel format: the encoder passes it through untouched
        embedding_output = self.embeddings(input_ids=ids, position_ids=position_ids, token_type_ids=token_type_ids, inputs_embeds=inputs_embeds)
        encoder_output = self.encoder(embedding_output, attention_mask=key_mask, return_attn_weights=return_attn_weights,
                                      return_hidden_states=return_hidden_states)
        last_hidden_state = encoder_output.last_hidden_state
        pooled_output = encoder_output.pooler_output
        if self.layernorm:
            last_hidden_state = self.layernorm(last_hidden_state)
        if self.pooler:
            pooled_output = self.pooler(last_hidden_state)
        return TransformerOutput(last_hidden_state=last_hidden_state, pooler_output=pooled_output,
                                 hidden_states=encoder_output.hidden_states, attentions=encoder_output.attentions)
