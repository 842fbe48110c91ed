"""Host-side mirror of torchmultimodal/modules/layers/text_embedding.py:13-104 (BERTTextEmbeddings).  Three gathers, the
sum and the LayerNorm are ONE kernel (csrc/rowops.hip: bert_embed_ln_kernel, wave per token)."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn, Tensor

from ... import ops
from ..._packing import PackedCache


class BERTTextEmbeddings(nn.Module):
    def __init__(
        self,
        hidden_size: int = 768,
        vocab_size: int = 30522,
        pad_token_id: int = 0,
        max_position_embeddings: int = 512,
        type_vocab_size: int = 2,
        layer_norm_eps: float = 1e-12,
        dropout: float = 0.0,
        offset_pos_ids: bool = False,
    ) -> None:
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab_size, hidden_size, pad_token_id)
        self.position_embeddings = nn.Embedding(max_position_embeddings, hidden_size)
        self.token_type_embeddings = nn.Embedding(type_vocab_size, hidden_size)
        self.layer_norm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)
        self.dropout = nn.Dropout(dropout)
        self.pad_token_id = pad_token_id
        self.offset_pos_ids = offset_pos_ids
        self._packed = PackedCache()

    def create_position_ids_from_input_ids(self, input_ids: Tensor) -> Tensor:
        """Non-padding tokens numbered from pad_token_id + 1, padding tokens = pad_token_id (reference :55-68); one row-scan kernel."""
        ids = input_ids if input_ids.dtype == torch.int64 else input_ids.to(torch.int64)
        return ops.offset_position_ids(ids if ids.is_contiguous() else ids.contiguous(), self.pad_token_id)

    def forward(
        self,
        input_ids: Optional[Tensor] = None,
        token_type_ids: Optional[Tensor] = None,
        position_ids: Optional[Tensor] = None,
        inputs_embeds: Optional[Tensor] = None,
    ) -> Tensor:
        if input_ids is None and inputs_embeds is None:
            raise ValueError("input_ids or inputs_embeds must not be None")
        if inputs_embeds is not None:
            # reference :95-101: the caller's embeddings take the place of the word-embedding rows (input_ids, if also given, then only feed the
            # position ids).  Inference form: the rows are read in place by the embedding kernel (row r of a [B*S, d] "table", id = r).
            if self.training and torch.is_grad_enabled() and (inputs_embeds.requires_grad or self.word_embeddings.weight.requires_grad):
                raise ops.MmamdError("inputs_embeds has no differentiable forward on the MI355X path (training takes input_ids); "
                                     "call .eval() / torch.no_grad() for inference")
            if inputs_embeds.dim() != 3 or inputs_embeds.dtype != torch.float32 or inputs_embeds.shape[-1] != self.word_embeddings.embedding_dim:
                raise ops.MmamdError(f"inputs_embeds must be fp32 [bsz, seq_len, {self.word_embeddings.embedding_dim}]")
            if self.training and self.dropout.p > 0:
                raise ops.MmamdError("embedding dropout applies on the differentiable (train mode, grad enabled) forward only: call .eval() for inference")
            B, S, d = inputs_embeds.shape
            if position_ids is None and self.offset_pos_ids:
                if input_ids is None:
                    raise ValueError("offset position ids are derived from input_ids")  # (the reference dereferences None here, :88-89)
                position_ids = self.create_position_ids_from_input_ids(input_ids)
            if position_ids is not None and tuple(position_ids.shape) != (B, S):
                position_ids = position_ids.expand(B, S).contiguous()
            rows = (inputs_embeds if inputs_embeds.is_contiguous() else inputs_embeds.contiguous()).view(B * S, d)
            row_ids = torch.arange(B * S, dtype=torch.int64, device=rows.device).view(B, S)  # index bookkeeping: table row of every token
            pk, f32 = self._packed.get, torch.float32
            x = ops.bert_embed_ln(row_ids, rows, pk(self.position_embeddings.weight, f32), pk(self.token_type_embeddings.weight, f32),
                                  pk(self.layer_norm.weight, f32), pk(self.layer_norm.bias, f32), self.layer_norm.eps, token_type_ids, position_ids)
            return x.view(B, S, -1)
        if self.offset_pos_ids and position_ids is None:
            position_ids = self.create_position_ids_from_input_ids(input_ids)  # reference :88-89
        B, S = input_ids.shape
        if position_ids is not None and position_ids.shape != input_ids.shape:
            position_ids = position_ids.expand(B, S).contiguous()
        ids = input_ids if input_ids.is_contiguous() else input_ids.contiguous()
        if self.training and torch.is_grad_enabled() and self.word_embeddings.weight.requires_grad:
            from ...models.flava._train import BertEmbedFn  # differentiable path

            from ..._autograd import dropout_train

            emb = BertEmbedFn.apply(ids, self.word_embeddings.weight, self.position_embeddings.weight, self.token_type_embeddings.weight,
                                    self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps, token_type_ids, position_ids,
                                    self.word_embeddings.padding_idx)
            return dropout_train(emb, self.dropout.p)  # reference :102-103: LayerNorm, then dropout
        if self.training and self.dropout.p > 0:
            raise ops.MmamdError("embedding dropout applies on the differentiable (train mode, grad enabled) forward only: call .eval() for inference")
        pk, f32 = self._packed.get, torch.float32
        x = ops.bert_embed_ln(input_ids if input_ids.is_contiguous() else input_ids.contiguous(),
                              pk(self.word_embeddings.weight, f32), pk(self.position_embeddings.weight, f32),
                              pk(self.token_type_embeddings.weight, f32), pk(self.layer_norm.weight, f32),
                              pk(self.layer_norm.bias, f32), self.layer_norm.eps, token_type_ids, position_ids)
        return x.view(B, S, -1)
