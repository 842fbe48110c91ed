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
    """Marker for "attend over all positions" (reference :13-57).  `attn_dropout` (dropout on the probabilities) is applied by the training
    forward of the enclosing encoder stack (the general attention kernels carry the Philox mask: _autograd.py); identity in eval mode."""

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
            out: Optional[Tensor] = None, qkv: Optional[Tensor] = None, att: Optional[Tensor] = None,
            head_mask: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
        """hn: bf16 [B*S, d] (already normalised).  Returns (fp32 [B*S, d] = output(attn) + residual, probs or None).  head_mask (reference
        attention.py:236-237): multiplied into the probabilities after the softmax -- the general attention kernel carries it."""
        d = self.query.in_features
        if self.key.in_features != d or self.d_qk != HEAD_DIM:
            raise ops.MmamdError("the MI355X attention kernel needs dim_q == dim_kv and 64-wide heads "
                                 f"(got dim_q={d}, dim_kv={self.key.in_features}, head dim {self.d_qk})")
        if not isinstance(self.attn, SelfAttention):
            raise ops.MmamdError(f"attn_module {type(self.attn).__name__} is not implemented on the MI355X path")
        if self.attn.attn_dropout > 0 and self.training:
            raise ops.MmamdError("this non-differentiable (stand-alone / inference) forward applies no dropout: call .eval(); training-time dropout runs inside the encoder / decoder stacks' differentiable forwards")
        bf, f32 = torch.bfloat16, torch.float32
        w = self._packed.get_cat([self.query.weight, self.key.weight, self.value.weight], bf)
        if self.query.bias is not None:
            b = self._packed.get_cat([self.query.bias, self.key.bias, self.value.bias], f32)
        else:
            b = None
        qkv = ops.gemm_bf16(hn, w, b, out=qkv)
        if head_mask is not None:
            att, probs = ops.attention_x_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, S, S, self.n_head, HEAD_DIM, ops.AttnMask(key_mask=key_mask),
                                             want_probs=want_probs, out=att, head_mask=head_mask_f32(head_mask))
        elif want_probs or key_mask is not None:
            att, probs = ops.attention_probs_fwd(qkv, B, S, self.n_head, key_mask, want_probs=want_probs, out=att)
        else:
            att, probs = ops.attention_fwd(qkv, B, S, self.n_head, causal=False, out=att), None
        y = ops.gemm_bf16(att, self._packed.get(self.output.weight, bf), self._packed.get(self.output.bias, f32),
                          residual=residual, out_dtype=f32, out=out)
        return y, probs

    def forward(self, q: Tensor, kv: Optional[Tensor] = None, return_attn_weights: bool = False, use_cache: bool = False,
                causal: bool = False, **attn_kwargs: Any) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        forbid_detached_forward(self, q, kv)
        am = attn_kwargs.get("attention_mask")
        hm = attn_kwargs.get("head_mask")
        if hm is not None:  # multiplicative post-softmax mask (:236-237): the general kernel, whatever the shapes
            return self._forward_general(q, kv, return_attn_weights, use_cache, causal, am, head_mask_f32(hm))
        pad_form = am is None or (am.dim() == 4 and am.shape[0] == q.shape[0] and am.shape[1] == 1 and am.shape[2] == 1) or (am.dim() == 2 and q.dim() == 3 and tuple(am.shape) == (q.shape[0], q.shape[1]) and q.shape[0] != q.shape[1])
        if q.dim() == 3 and (kv is None or kv is q) and not use_cache and not self.cache and pad_form:
            # self-attention over [b, seq, c] without a cache (what FLAVA's layers do): one packed in-projection, the flash-style kernels
            # (`causal` only steers the CACHE in the reference, :159-176: masking comes from attention_mask)
            B, S, d = q.shape
            qc = q if q.is_contiguous() else q.contiguous()
            km = key_mask_from_attention_mask(attn_kwargs.get("attention_mask"), B, S)
            y, probs = self.run(ops.convert(qc.view(B * S, d), torch.bfloat16), B, S, km, return_attn_weights, None)
            y = y.view(B, S, d)
            return (y, probs) if return_attn_weights else y
        return self._forward_general(q, kv, return_attn_weights, use_cache, causal, attn_kwargs.get("attention_mask"))

    def _forward_general(self, q: Tensor, kv: Optional[Tensor], return_attn_weights: bool, use_cache: bool, causal: bool,
                         attention_mask: Optional[Tensor], head_mask: Optional[Tensor] = None) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        """Cross-attention (kv), n-dimensional token grids, and the key / value cache of the reference's decoding loop (:150-176): q, k, v are
        projected separately (k and v by one stacked GEMM), the general attention kernel (mmamd_attention_x_fwd: Sq != Sk, 64- / 96-wide heads,
        key-padding or [Sq, Sk] masks) attends, `self.cache` holds {"k", "v"} in the reference's [b, n_head, seq, c] shape as bf16 views of
        token-major buffers.  `causal` steers the cache exactly like the reference: with it, new keys are appended to the cached ones; without
        it a filled cache REPLACES the projections of `kv` (cross-attention to a fixed memory)."""
        if not isinstance(self.attn, SelfAttention):
            raise ops.MmamdError(f"attn_module {type(self.attn).__name__} is not implemented on the MI355X path")
        if self.attn.attn_dropout > 0 and self.training:
            raise ops.MmamdError("this non-differentiable (stand-alone / inference) forward applies no dropout: call .eval(); training-time dropout runs inside the encoder / decoder stacks' differentiable forwards")
        dq, H = self.query.out_features, self.n_head
        hd = dq // H
        if hd not in (64, 96):
            raise ops.MmamdError(f"the MI355X attention kernels are built for 64- and 96-wide heads, got {hd}")
        bf, f32, pc = torch.bfloat16, torch.float32, self._packed
        B = q.shape[0]
        q_shape = q.shape
        src = q if kv is None else kv
        Sq = q.numel() // (B * q.shape[-1])
        Sn = src.numel() // (B * src.shape[-1])
        qc = (q if q.is_contiguous() else q.contiguous()).view(B * Sq, q.shape[-1])
        has_b = self.query.bias is not None
        qp = ops.gemm_bf16(ops.convert(qc, bf), pc.get(self.query.weight, bf), pc.get(self.query.bias, f32) if has_b else None)
        k_tok = v_tok = None
        if causal or not self.cache:
            sc = (src if src.is_contiguous() else src.contiguous()).view(B * Sn, src.shape[-1])
            wkv = pc.get_cat([self.key.weight, self.value.weight], bf)
            bkv = pc.get_cat([self.key.bias, self.value.bias], f32) if has_b else None
            kvp = ops.gemm_bf16(ops.convert(sc, bf) if sc is not qc else ops.convert(qc, bf), wkv, bkv)
            k_tok, v_tok = kvp[:, :dq].unflatten(0, (B, Sn)), kvp[:, dq:].unflatten(0, (B, Sn))  # [B, Sn, dq] column-slice views

        def as_heads(tok):  # [B, S, dq] token-major -> the reference's [b, n_head, S, c] (a view)
            return tok.view(B, tok.shape[1], H, hd).transpose(1, 2)

        def as_tokens(t4):  # the inverse, for cache entries (free when they came from as_heads)
            return t4.transpose(1, 2).reshape(B, t4.shape[2], dq)

        if use_cache:
            if not self.cache:
                self.cache = dict(k=as_heads(k_tok.contiguous()), v=as_heads(v_tok.contiguous()))  # (.contiguous(): the reference clones)
            elif causal:  # append the present keys / values to the past ones (data movement)
                self.cache["k"] = as_heads(torch.cat([as_tokens(self.cache["k"]), k_tok], dim=1))
                self.cache["v"] = as_heads(torch.cat([as_tokens(self.cache["v"]), v_tok], dim=1))
            k_tok, v_tok = as_tokens(self.cache["k"]), as_tokens(self.cache["v"])
        Sk = k_tok.shape[1]
        k2 = k_tok.reshape(B * Sk, dq) if k_tok.is_contiguous() else k_tok.flatten(0, 1)
        v2 = v_tok.reshape(B * Sk, dq) if v_tok.is_contiguous() else v_tok.flatten(0, 1)
        mask = _general_mask(attention_mask, B, Sq, Sk)
        att, probs = ops.attention_x_fwd(qp, k2, v2, B, Sq, Sk, H, hd, mask, want_probs=return_attn_weights, head_mask=head_mask)
        y = ops.gemm_bf16(att, pc.get(self.output.weight, bf), pc.get(self.output.bias, f32), out_dtype=f32)
        y = y.view(*q_shape[:-1], dq)
        return (y, probs) if return_attn_weights else y


def head_mask_f32(head_mask: Tensor) -> Tensor:
    """The reference's head_mask (any dtype, broadcastable to [b, h, q, k]) as the contiguous fp32 device tensor the kernel strides over."""
    if not head_mask.is_cuda:
        raise ops.MmamdError("head_mask must live on the HIP device")
    hm = head_mask.detach().to(torch.float32)
    return hm if hm.is_contiguous() else hm.contiguous()


def _general_mask(attention_mask: Optional[Tensor], B: int, Sq: int, Sk: int) -> ops.AttnMask:
    """The reference's "0 = do not attend" masks, broadcastable to [b, h, q, k] (:203-206, utils/attention.py): key-padding forms ([B,Sk],
    [B,1,1,Sk]) become a key mask, head-independent [Sq,Sk] / [B,1,Sq,Sk] / [1,1,Sq,Sk] forms a full uint8 mask; per-head masks raise."""
    if attention_mask is None:
        return ops.AttnMask()
    m = attention_mask
    if m.dim() > 4:
        raise ops.MmamdError(f"attention_mask of shape {tuple(m.shape)}: at most 4 dimensions ([b, h, q, k] broadcasting)")
    m = m.reshape((1,) * (4 - m.dim()) + tuple(m.shape))  # torch broadcasting aligns from the right
    if m.shape[1] != 1:
        raise ops.MmamdError("per-head attention masks are not implemented on the MI355X path")
    if m.shape[0] not in (1, B) or m.shape[3] != Sk or m.shape[2] not in (1, Sq):
        raise ops.MmamdError(f"attention_mask of shape {tuple(attention_mask.shape)} does not broadcast to [{B}, h, {Sq}, {Sk}]")
    flags = (m[:, 0] != 0).to(torch.uint8)  # mask plumbing: 0 / 1 flags
    if m.shape[2] == 1 and Sq != 1:
        return ops.AttnMask(key_mask=flags[:, 0].expand(B, Sk).contiguous())
    return ops.AttnMask(full=flags.expand(flags.shape[0], Sq, Sk).contiguous())
