"""Host-side mirror of torchmultimodal/modules/layers/attention.py:60-241 (MultiHeadAttention + SelfAttention), the attention
block of FLAVA's encoder layers.  Parameter names (`query`, `key`, `value`, `output`) and construction order match the
reference.  The self-attention forward is three kernels: ONE in-projection GEMM over the stacked [3d, d] weight, the
flash-style MFMA attention (csrc/attention.hip: probabilities + key-padding mask variant), the output GEMM.
"""
from __future__ import annotations

from typing import Any, Optional, Tuple, Union

import torch
from torch import nn, Tensor

from ... import ops
from ..._autograd import forbid_detached_forward
from ..._packing import PackedCache

HEAD_DIM = 64


class SelfAttention(nn.Module):
    """Marker for "attend over all positions" (reference :13-57).  Dropout on the probabilities is not implemented on the
    MI355X path (every FLAVA factory builds it with 0.0)."""

    def __init__(self, attn_dropout: float = 0.0) -> None:
        super().__init__()
        self.attn_dropout = attn_dropout


def key_mask_from_attention_mask(attention_mask: Optional[Tensor], B: int, S: int) -> Optional[Tensor]:
    """The reference's masks are "0 = do not attend" tensors broadcastable to [B,H,Sq,Sk] (utils/attention.py:13-52).  The
    kernel takes key-padding masks: [B,S], [B,1,1,S] (what BERTTextEncoder builds) — anything query- or head-dependent
    raises."""
    if attention_mask is None:
        return None
    m = attention_mask
    if m.dim() == 4 and m.shape[1] == 1 and m.shape[2] == 1:
        m = m.reshape(m.shape[0], m.shape[3])
    if m.dim() != 2 or tuple(m.shape) != (B, S):
        raise ops.MmamdError(f"attention_mask of shape {tuple(attention_mask.shape)}: only key-padding masks ([B,S] or "
                             "[B,1,1,S], 0 = masked) are implemented on the MI355X path")
    if m.dtype == torch.uint8 and getattr(m, "_mmamd_key_mask", False):
        return m
    return ops.key_mask(m if m.is_contiguous() else m.contiguous())


class MultiHeadAttention(nn.Module):
    def __init__(self, dim_q: int, dim_kv: int, n_head: int, attn_module: nn.Module = SelfAttention(), add_bias: bool = True) -> None:
        super().__init__()
        if dim_q % n_head != 0 or dim_kv % n_head != 0:
            raise ValueError("The hidden size of q, k, v must be a multiple of the number of attention heads.")
        self.d_qk = dim_q // n_head
        self.d_v = dim_kv // n_head
        self.n_head = n_head
        self.query = nn.Linear(dim_q, dim_q, bias=add_bias)
        self.key = nn.Linear(dim_kv, dim_q, bias=add_bias)
        self.value = nn.Linear(dim_kv, dim_q, bias=add_bias)
        self.output = nn.Linear(dim_q, dim_q, bias=True)
        self.attn = attn_module
        self.cache = None
        self._packed = PackedCache()

    def run(self, hn: Tensor, B: int, S: int, key_mask: Optional[Tensor], want_probs: bool, residual: Optional[Tensor],
            out: Optional[Tensor] = None, qkv: Optional[Tensor] = None, att: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
        """hn: bf16 [B*S, d] (already normalised).  Returns (fp32 [B*S, d] = output(attn) + residual, probs or None)."""
        d = self.query.in_features
        if self.key.in_features != d or self.d_qk != HEAD_DIM:
            raise ops.MmamdError("the MI355X attention kernel needs dim_q == dim_kv and 64-wide heads "
                                 f"(got dim_q={d}, dim_kv={self.key.in_features}, head dim {self.d_qk})")
        if not isinstance(self.attn, SelfAttention):
            raise ops.MmamdError(f"attn_module {type(self.attn).__name__} is not implemented on the MI355X path")
        if self.attn.attn_dropout > 0 and self.training:
            raise ops.MmamdError("attention dropout > 0 in training mode is not implemented on the MI355X path")
        bf, f32 = torch.bfloat16, torch.float32
        w = self._packed.get_cat([self.query.weight, self.key.weight, self.value.weight], bf)
        if self.query.bias is not None:
            b = self._packed.get_cat([self.query.bias, self.key.bias, self.value.bias], f32)
        else:
            b = None
        qkv = ops.gemm_bf16(hn, w, b, out=qkv)
        if want_probs or key_mask is not None:
            att, probs = ops.attention_probs_fwd(qkv, B, S, self.n_head, key_mask, want_probs=want_probs, out=att)
        else:
            att, probs = ops.attention_fwd(qkv, B, S, self.n_head, causal=False, out=att), None
        y = ops.gemm_bf16(att, self._packed.get(self.output.weight, bf), self._packed.get(self.output.bias, f32),
                          residual=residual, out_dtype=f32, out=out)
        return y, probs

    def forward(self, q: Tensor, kv: Optional[Tensor] = None, return_attn_weights: bool = False, use_cache: bool = False,
                causal: bool = False, **attn_kwargs: Any) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        if kv is not None and kv is not q:
            raise ops.MmamdError("cross-attention is not on the MI355X contrastive path")
        if use_cache or causal or attn_kwargs.get("head_mask") is not None:
            raise ops.MmamdError("use_cache / causal / head_mask are not implemented on the MI355X path")
        if q.dim() != 3:
            raise ops.MmamdError("MultiHeadAttention on the MI355X path takes [b, seq, c] inputs")
        forbid_detached_forward(self, q)
        B, S, d = q.shape
        qc = q if q.is_contiguous() else q.contiguous()
        km = key_mask_from_attention_mask(attn_kwargs.get("attention_mask"), B, S)
        y, probs = self.run(ops.convert(qc.view(B * S, d), torch.bfloat16), B, S, km, return_attn_weights, None)
        y = y.view(B, S, d)
        return (y, probs) if return_attn_weights else y
