"""Host-side mirror of torchmultimodal/models/coca/text_decoder.py:17-252 (CoCaTextEmbeddings, CoCaTextDecoder).

Token gather + CLS row + position add is one kernel (coca_text_embed_kernel); the padding-aware causal mask of
`build_mask` is produced directly as the uint8 [B, S, S] mask the attention kernel reads (coca_text_mask_kernel)."""

from typing import Callable, Optional, Tuple

import torch
from torch import nn, Tensor

from ... import _torch_ops, ops
from ..._packing import PackedCache
from ...modules.layers.transformer import TransformerDecoder
from ...utils.attention import get_causal_attention_mask


_torch_ops.try_load()


class CoCaTextEmbedFn(torch.autograd.Function):
    """token_embeddings[ids] (+ CLS row appended) + position_embeddings, differentiable (models/coca/text_decoder.py:67-88)."""

    @staticmethod
    def forward(ctx, ids, table, pos, cls, padding_idx):
        f32 = torch.float32
        x = ops.coca_text_embed(ids, table.detach().contiguous(), pos.detach().contiguous(), cls.detach().contiguous() if cls is not None else None)
        B, S = ids.shape
        T = S + (1 if cls is not None else 0)
        ctx.save_for_backward(ids)
        ctx.meta = (tuple(table.shape), tuple(pos.shape), cls is not None, padding_idx, T)
        return x.view(B, T, -1)

    @staticmethod
    def backward(ctx, dx):
        (ids,) = ctx.saved_tensors
        tshape, pshape, has_cls, padding_idx, T = ctx.meta
        f32 = torch.float32
        B, S = ids.shape
        d = tshape[1]
        dxc = dx.detach().contiguous().view(B * T, d)
        dpos_used = ops.colsum(dxc.view(B, T * d)).view(T, d)     # position t is shared by the B samples
        dpos = torch.zeros(pshape, dtype=f32, device=dx.device)
        dpos[:T].copy_(dpos_used)
        dcls = dpos_used[S].clone() if has_cls else None          # the CLS row is cls + pos[S] for every sample
        rows = (torch.arange(B * T, device=dx.device, dtype=torch.int32).view(B, T)[:, :S]).reshape(-1).contiguous()
        dtok = ops.gather_rows(dxc, d, rows, d, f32)               # gradients of the token rows, [B*S, d]
        dtable = torch.zeros(tshape, dtype=f32, device=dx.device)  # memset; rows collide -> fp32 atomics
        ops.scatter_add_rows_(dtable, ids.reshape(-1).contiguous(), dtok)
        if padding_idx is not None:
            dtable[padding_idx].zero_()
        return None, dtable, dpos, dcls, None


class CoCaTextEmbeddings(nn.Module):
    def __init__(self, vocab_size: int, num_positions: int, embedding_dim: int, pad_idx: Optional[int] = 0, embed_cls: bool = True):
        super().__init__()
        self.num_positions = num_positions
        if embed_cls:
            self.cls_embedding = nn.Parameter(torch.empty(embedding_dim))
        else:
            self.cls_embedding = None
        self.token_embeddings = nn.Embedding(vocab_size, embedding_dim, pad_idx)
        self.position_embeddings = nn.Parameter(torch.empty(num_positions, embedding_dim))
        self.init_parameters()
        self._packed = PackedCache()

    def init_parameters(self) -> None:
        nn.init.normal_(self.token_embeddings.weight, std=0.02)
        nn.init.normal_(self.position_embeddings, std=0.01)
        if self.cls_embedding is not None:
            nn.init.constant_(self.cls_embedding, 0.01)

    def forward(self, input_ids: Tensor) -> Tensor:
        if torch.jit.is_scripting():
            return self._forward_ops(input_ids)
        else:
            return self._forward_host(input_ids)

    def _forward_ops(self, input_ids: Tensor) -> Tensor:
        """The forward through the dispatcher op (torch.ops.mmamd.coca_text_embed) — what torch.jit.script / torch.compile see."""
        B = input_ids.size(0)
        cls: Optional[Tensor] = None
        T = self.num_positions
        if self.cls_embedding is not None:
            cls = self.cls_embedding
            T = self.num_positions - 1
        assert input_ids.size(1) == T
        x = torch.ops.mmamd.coca_text_embed(input_ids.contiguous(), self.token_embeddings.weight, self.position_embeddings, cls)
        return x.view(B, -1, x.size(1))

    @torch.jit.unused
    def _forward_host(self, input_ids: Tensor) -> Tensor:
        assert input_ids.shape[1] == (self.num_positions if self.cls_embedding is None else self.num_positions - 1)
        if torch.compiler.is_compiling() and not (self.training and torch.is_grad_enabled() and self.token_embeddings.weight.requires_grad):
            return self._forward_ops(input_ids)
        pk, f32 = self._packed.get, torch.float32
        ids = input_ids if input_ids.is_contiguous() else input_ids.contiguous()
        if self.training and torch.is_grad_enabled() and self.token_embeddings.weight.requires_grad:
            return CoCaTextEmbedFn.apply(ids, self.token_embeddings.weight, self.position_embeddings, self.cls_embedding,
                                         self.token_embeddings.padding_idx)
        x = ops.coca_text_embed(ids, pk(self.token_embeddings.weight, f32), pk(self.position_embeddings, f32),
                                pk(self.cls_embedding, f32) if self.cls_embedding is not None else None)
        return x.view(ids.shape[0], -1, x.shape[-1])


class CoCaTextDecoder(nn.Module):
    def __init__(self, vocab_size: int, num_positions: int, embedding_dim: int, n_layer: int, n_head: int, dim_feedforward: int,
                 output_dim: int, pad_idx: Optional[int] = 0, embed_cls: bool = True, dropout: float = 0.0,
                 activation: Callable[..., nn.Module] = nn.GELU, layer_norm_eps: float = 1e-5, norm_first: bool = True,
                 final_layer_norm_eps: Optional[float] = 1e-5):
        super().__init__()
        self.pad_idx = pad_idx
        self.embed_cls = embed_cls
        self.num_positions = num_positions
        self.embeddings = CoCaTextEmbeddings(vocab_size=vocab_size, num_positions=num_positions, embedding_dim=embedding_dim,
                                             pad_idx=pad_idx, embed_cls=embed_cls)
        self.transformer_decoder = TransformerDecoder(n_layer=n_layer, d_model=embedding_dim, n_head=n_head,
                                                      dim_feedforward=dim_feedforward, dropout=dropout, activation=activation,
                                                      layer_norm_eps=layer_norm_eps, norm_first=norm_first, use_cross_attention=False)
        if final_layer_norm_eps is not None:
            self.ln_final = nn.LayerNorm(normalized_shape=embedding_dim, eps=final_layer_norm_eps)
        self._ln_final_eps: float = final_layer_norm_eps if final_layer_norm_eps is not None else 0.0
        self.text_projection = nn.Linear(embedding_dim, output_dim, bias=False)
        self.register_buffer("causal_mask", get_causal_attention_mask(num_positions).to(dtype=torch.bool), persistent=False)
        self.init_parameters(embedding_dim, n_layer)
        self._packed = PackedCache()

    def init_parameters(self, embedding_dim: int, n_layer: int) -> None:
        attn_std = embedding_dim**-0.5
        proj_std = (2 * embedding_dim * n_layer) ** -0.5
        fc_std = (2 * embedding_dim) ** -0.5
        for layer in self.transformer_decoder.layer:
            nn.init.normal_(layer.attention.q_proj.weight, std=attn_std)
            nn.init.normal_(layer.attention.k_proj.weight, std=attn_std)
            nn.init.normal_(layer.attention.v_proj.weight, std=attn_std)
            nn.init.normal_(layer.attention.output_proj.weight, std=proj_std)
            nn.init.normal_(layer.feedforward.model[0].weight, std=fc_std)
            nn.init.normal_(layer.feedforward.model[2].weight, std=proj_std)
        nn.init.normal_(self.text_projection.weight, std=embedding_dim**0.5)

    def build_mask(self, input_ids: Tensor, padding_mask: Optional[Tensor] = None):
        """Kernel-format mask: plain causal when there is no CLS / pad handling, else the uint8 [B, S+1, S+1] mask whose CLS
        row hides padded tokens (reference :178-194)."""
        if not self.embed_cls or self.pad_idx is None:
            return ops.AttnMask(causal=True)
        # The reference's mask is causal in every row; only the CLS query (the last row) additionally hides key j >= 1 when token j - 1 is padding
        # (F.pad(..., (1, 0, S, 0)) shifts the padding mask by one column and fills the rows above with ones).  The attention kernels take exactly
        # that as flags -- causal + a key mask [B, S + 1] that binds the last query row only -- instead of reading a [B, S + 1, S + 1] byte tensor
        # per score (r04: decoder self-attention 122 -> 76 us with 32-bit mask loads, and no mask loads at all this way).  The dense form stays what
        # the scripted forward builds (mmamd_coca_text_mask) and what `dense_mask()` returns.
        if padding_mask is None:
            km = ops.key_mask(input_ids if input_ids.is_contiguous() else input_ids.contiguous(), pad_id=self.pad_idx)
        else:
            km = ops.key_mask(padding_mask if padding_mask.is_contiguous() else padding_mask.contiguous())
        km = torch.cat([torch.ones((km.shape[0], 1), dtype=torch.uint8, device=km.device), km], dim=1)  # key 0 = the first token: always visible (mask plumbing)
        return ops.AttnMask(causal=True, key_mask=km, key_mask_last_row=True)

    def dense_mask(self, input_ids: Tensor, padding_mask: Optional[Tensor] = None) -> Tensor:
        """The reference's build_mask as a uint8 [B, S + 1, S + 1] tensor (0 = do not attend)."""
        if padding_mask is None:
            return ops.coca_text_mask(input_ids if input_ids.is_contiguous() else input_ids.contiguous(), pad_id=self.pad_idx)
        return ops.coca_text_mask(padding_mask if padding_mask.is_contiguous() else padding_mask.contiguous())

    def forward(self, input_ids: Tensor, padding_mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        if torch.jit.is_scripting():
            return self._forward_ops(input_ids, padding_mask)
        else:
            return self._forward_host(input_ids, padding_mask)

    def _forward_ops(self, input_ids: Tensor, padding_mask: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
        """The forward through the dispatcher ops — what torch.jit.script / torch.compile see (inference; embed_cls decoders)."""
        if not self.embed_cls:
            raise RuntimeError("CoCaTextDecoder(embed_cls=False) is not implemented on the MI355X path")
        if input_ids.size(1) == self.num_positions:
            input_ids = input_ids[:, :-1]
        if padding_mask is not None:
            if padding_mask.size(1) == self.num_positions:
                padding_mask = padding_mask[:, :-1]
        assert input_ids.size(1) == self.num_positions - 1
        ids = input_ids.contiguous()
        embeddings = self.embeddings(ids)
        full: Optional[Tensor] = None
        pad_idx = self.pad_idx
        if pad_idx is not None:  # the padding-aware causal mask of build_mask (uint8 [B, S+1, S+1]); else plain causal
            if padding_mask is None:
                full = torch.ops.mmamd.coca_text_mask(ids, True, pad_idx)
            else:
                full = torch.ops.mmamd.coca_text_mask(padding_mask.contiguous(), False, 0)
        hidden_states = self.transformer_decoder._forward_ops(embeddings, None, full is None, full, False).last_hidden_state
        assert hidden_states is not None, "hidden states must not be None"
        B, S, d = hidden_states.size(0), hidden_states.size(1), hidden_states.size(2)
        tokens = hidden_states[:, :-1]
        pooled = hidden_states[:, S - 1].contiguous()  # the B CLS rows (data movement)
        if hasattr(self, "ln_final"):
            pooled = torch.ops.mmamd.layernorm(pooled, self.ln_final.weight, self.ln_final.bias, self._ln_final_eps, 0)
        pooled = torch.ops.mmamd.rows_linear_f32(pooled, self.text_projection.weight, None)
        return pooled, tokens

    @torch.jit.unused
    def _forward_host(self, input_ids: Tensor, padding_mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        if torch.compiler.is_compiling() and self.embed_cls and not torch.is_grad_enabled():
            return self._forward_ops(input_ids, padding_mask)
        if self.embed_cls:
            if input_ids.shape[1] == self.num_positions:
                input_ids = input_ids[:, :-1]
            if padding_mask is not None and padding_mask.shape[1] == self.num_positions:
                padding_mask = padding_mask[:, :-1]
        target_shape = self.num_positions - 1 if self.embed_cls else self.num_positions
        assert input_ids.shape[1] == target_shape, f"{input_ids.shape} doesn't match ({target_shape},*)"
        input_ids = input_ids if input_ids.is_contiguous() else input_ids.contiguous()
        embeddings = self.embeddings(input_ids)
        mask = self.build_mask(input_ids, padding_mask)
        hidden_states = self.transformer_decoder(embeddings, attention_mask=mask).last_hidden_state
        assert hidden_states is not None, "hidden states must not be None"
        pk, f32 = self._packed.get, torch.float32
        B, S, d = hidden_states.shape
        if self.embed_cls and torch.is_grad_enabled() and hidden_states.requires_grad:
            from ..clip._train import PooledHeadFn  # differentiable: LayerNorm on the CLS rows + projection (no bias)

            if getattr(self, "ln_final", None) is None or self.text_projection is None:
                raise ops.MmamdError("training on the MI355X path: CoCaTextDecoder needs ln_final and text_projection")
            rows = torch.arange(S - 1, B * S, S, dtype=torch.int64, device=hidden_states.device)
            pooled = PooledHeadFn.apply(hidden_states.reshape(B * S, d), rows, self.ln_final.weight, self.ln_final.bias,
                                        self.text_projection.weight, self.ln_final.eps, True)
            return pooled, hidden_states[:, :-1]
        if self.embed_cls:
            tokens = hidden_states[:, :-1]
            # pooled = text_projection(ln_final(hidden[:, -1])): gather the B CLS rows, LN, exact-fp32 MFMA projection
            pooled = ops.gather_rows(hidden_states.view(B * S, d), d, _last_row_index(B, S, hidden_states.device), d, f32)
            if getattr(self, "ln_final", None) is not None:
                pooled = ops.layernorm(pooled, pk(self.ln_final.weight, f32), pk(self.ln_final.bias, f32), self.ln_final.eps, out_dtype=f32)
            if self.text_projection is not None:
                pooled = ops.rows_linear_f32(pooled, d, B, pk(self.text_projection.weight, f32), None)
        else:
            raise ops.MmamdError("CoCaTextDecoder(embed_cls=False) is not implemented on the MI355X path")
        return pooled, tokens


_LAST_ROWS = {}


def _last_row_index(B: int, S: int, device: torch.device) -> Tensor:
    """Row numbers of every sample's last position in a [B*S, d] matrix (index bookkeeping, cached)."""
    key = (B, S, device)
    if key not in _LAST_ROWS:
        _LAST_ROWS[key] = torch.arange(S - 1, B * S, S, dtype=torch.int32, device=device)
    return _LAST_ROWS[key]
