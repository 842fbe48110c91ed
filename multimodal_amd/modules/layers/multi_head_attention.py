"""Host-side mirror of torchmultimodal/modules/layers/multi_head_attention.py:19-180 (MultiHeadSelfAttention,
MultiHeadAttentionWithCache, MHAWithCacheOutput) — the attention blocks of CoCa's ViT, decoders and attention pooler.

Same constructors and parameter names (`input_proj` / `output_proj`; `q_proj`, `k_proj`, `v_proj`, `output_proj`).  Forward on
the MI355X: one GEMM for the stacked projections (q|k|v for self-attention, k|v of the encoder states for cross-attention),
the general MFMA attention kernel (csrc/attention.hip: attention_x_kernel — Sq != Sk, 64- or 96-wide heads, causal / padding /
full boolean masks, batch-shared queries), one GEMM for the output projection with the residual add in its epilogue.
"""

from typing import NamedTuple, Optional, Tuple, Union

import torch
from torch import nn, Tensor

from ... import _torch_ops, ops
from ..._autograd import forbid_detached_forward
from ..._packing import PackedCache
from ...ops import AttnMask


_torch_ops.try_load()


class MHAWithCacheOutput(NamedTuple):
    attn_output: Tensor
    past_key_value: Tuple[Tensor, Tensor]


def to_attn_mask(attn_mask: Optional[Tensor], is_causal: bool, B: int, Sq: int, Sk: int) -> AttnMask:
    """Reference-style masks -> kernel masks.  Boolean masks (True = take part) of shape [Sq,Sk], [B,Sq,Sk], [B,1,Sq,Sk] or
    [1,1,Sq,Sk] become a uint8 full mask; additive float masks and per-head masks are not implemented."""
    if isinstance(attn_mask, AttnMask):
        return attn_mask
    if attn_mask is None:
        return AttnMask(causal=is_causal)
    if is_causal:
        raise ops.MmamdError("attn_mask must be None when is_causal=True")
    m = attn_mask
    if m.dtype not in (torch.bool, torch.uint8):
        raise ops.MmamdError(f"attention masks on the MI355X path are boolean (True = attend); got {m.dtype} (additive float "
                             "masks are not implemented)")
    if m.dim() == 4:
        if m.shape[1] != 1:
            raise ops.MmamdError("per-head attention masks are not implemented on the MI355X path")
        m = m[:, 0]
    if m.dim() == 2:
        m = m[None]
    if m.dim() != 3 or tuple(m.shape[-2:]) != (Sq, Sk) or m.shape[0] not in (1, B):
        raise ops.MmamdError(f"attention mask of shape {tuple(attn_mask.shape)} does not broadcast to [{B}, 1, {Sq}, {Sk}]")
    m = m if m.is_contiguous() else m.contiguous()
    return AttnMask(full=ops.key_mask(m))


def _check_heads(embed_dim: int, num_heads: int) -> int:
    hd = embed_dim // num_heads
    if hd * num_heads != embed_dim or hd not in (64, 96):
        raise ops.MmamdError(f"the MI355X attention kernels are built for 64- and 96-wide heads, got {embed_dim}/{num_heads}")
    return hd


class MultiHeadSelfAttention(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, dropout: float = 0.0):
        super().__init__()
        self.input_proj = nn.Linear(embed_dim, 3 * embed_dim)
        self.output_proj = nn.Linear(embed_dim, embed_dim)
        self.num_heads = num_heads
        self.dropout = dropout
        self._packed = PackedCache()

    @torch.jit.unused
    def run(self, hn: Tensor, B: int, S: int, mask: AttnMask, residual: Optional[Tensor], out: Optional[Tensor] = None) -> Tensor:
        """hn: bf16 [B*S, d] -> fp32 [B*S, d] = output_proj(attention) (+ residual)."""
        if self.training and self.dropout > 0:
            raise ops.MmamdError("this non-differentiable (stand-alone / inference) forward applies no dropout: call .eval(); training-time dropout runs inside the encoder / decoder stacks' differentiable forwards")
        d = self.output_proj.in_features
        hd = _check_heads(d, self.num_heads)
        pk, bf, f32 = self._packed.get, torch.bfloat16, torch.float32
        qkv = ops.gemm_bf16(hn, pk(self.input_proj.weight, bf), pk(self.input_proj.bias, f32))
        if mask.empty and hd == 64:
            att = ops.attention_fwd(qkv, B, S, self.num_heads, causal=False)
        elif mask.causal and mask.key_mask is None and mask.full is None and hd == 64:
            att = ops.attention_fwd(qkv, B, S, self.num_heads, causal=True)
        else:
            att, _ = ops.attention_x_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, S, S, self.num_heads, hd, mask)
        return ops.gemm_bf16(att, pk(self.output_proj.weight, bf), pk(self.output_proj.bias, f32), residual=residual,
                             out_dtype=f32, out=out)

    def forward(self, query: Tensor, attn_mask: Optional[Tensor] = None, is_causal: bool = False) -> Tensor:
        if torch.jit.is_scripting():
            return self._forward_ops(query, attn_mask, is_causal)
        else:
            return self._forward_host(query, attn_mask, is_causal)

    def _forward_ops(self, query: Tensor, attn_mask: Optional[Tensor], is_causal: bool) -> Tensor:
        """The forward through the dispatcher ops (torch.ops.mmamd.*, csrc/torch_ops.cpp) — what torch.jit.script sees (reference:
        tests/modules/layers/test_multi_head_attention.py:50-57 scripts this module and calls it without a mask).  64-wide heads,
        no mask other than is_causal; everything else needs the eager forward."""
        if attn_mask is not None:
            raise RuntimeError("scripted MultiHeadSelfAttention on the MI355X path takes no attn_mask (use is_causal, or the eager forward)")
        if query.dim() != 3:
            raise RuntimeError("MultiHeadSelfAttention takes bsz x seq_len x embed_dim inputs")
        B, S, d = query.size(0), query.size(1), query.size(2)
        if d != 64 * self.num_heads and d != 96 * self.num_heads:
            raise RuntimeError("the MI355X attention kernels are built for 64- and 96-wide heads")
        x = torch.ops.mmamd.convert(query.contiguous().view(B * S, d), 1)
        return self._run_ops(x, B, S, is_causal, None).view(B, S, d)

    def _run_ops(self, hn: Tensor, B: int, S: int, is_causal: bool, residual: Optional[Tensor]) -> Tensor:
        """run() through the dispatcher ops: hn bf16 [B*S, d] -> fp32 [B*S, d] = output_proj(attention) (+ residual)."""
        d = self.output_proj.in_features
        H = self.num_heads
        hd = d // H
        qkv = torch.ops.mmamd.gemm_bf16(hn, self.input_proj.weight, self.input_proj.bias, None, 0, 1)
        if hd == 64:
            att = torch.ops.mmamd.attn_fwd(qkv, B, S, H, is_causal)
        else:
            att = torch.ops.mmamd.attn_x(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, S, S, H, hd, is_causal, None, None, False)
        return torch.ops.mmamd.gemm_bf16(att, self.output_proj.weight, self.output_proj.bias, residual, 0, 0)

    @torch.jit.unused
    def _forward_host(self, query: Tensor, attn_mask: Optional[Tensor] = None, is_causal: bool = False) -> Tensor:
        if query.dim() != 3:
            raise ops.MmamdError("MultiHeadSelfAttention takes bsz x seq_len x embed_dim inputs")
        forbid_detached_forward(self, query)
        B, S, d = query.shape
        qc = query if query.is_contiguous() else query.contiguous()
        mask = to_attn_mask(attn_mask, is_causal, B, S, S)
        return self.run(ops.convert(qc.view(B * S, d), torch.bfloat16), B, S, mask, None).view(B, S, d)


class MultiHeadAttentionWithCache(nn.Module):
    def __init__(self, dim_q: int, dim_kv: int, num_heads: int, dropout: float = 0.0, add_bias: bool = True) -> None:
        super().__init__()
        self.num_heads = num_heads
        self.q_proj = nn.Linear(dim_q, dim_q, bias=add_bias)
        self.k_proj = nn.Linear(dim_kv, dim_q, bias=add_bias)
        self.v_proj = nn.Linear(dim_kv, dim_q, bias=add_bias)
        self.output_proj = nn.Linear(dim_q, dim_q)
        self.dropout = dropout
        self._packed = PackedCache()

    def run(self, q_in: Tensor, kv_in: Optional[Tensor], B: int, Sq: int, Sk: int, mask: AttnMask, residual: Optional[Tensor],
            shared_q: bool = False, out: Optional[Tensor] = None, past: Optional[Tuple[Tensor, Tensor]] = None,
            use_cache: bool = False):
        """q_in: bf16 [B*Sq, dq] (or [Sq, dq] when shared_q); kv_in: bf16 [B*Sk, dkv], None = self-attention over q_in.
        Returns fp32 [B*Sq, dq] = output_proj(attention) (+ residual).
        Incremental decoding (reference :158-179): `past` = (key, value) [B, H, Sp, hd] of the earlier positions is prepended to the
        Sk new keys / values (mask then spans Sp + Sk keys); with use_cache the result is (output, (key, value)) — the new cache in the
        reference's [B, H, Sp + Sk, hd] shape, held in bf16 as views of one token-major buffer (feeding it back costs no conversion)."""
        if self.training and self.dropout > 0:
            raise ops.MmamdError("this non-differentiable (stand-alone / inference) forward applies no dropout: call .eval(); training-time dropout runs inside the encoder / decoder stacks' differentiable forwards")
        dq = self.q_proj.out_features
        hd = _check_heads(dq, self.num_heads)
        pc, bf, f32 = self._packed, torch.bfloat16, torch.float32
        has_b = self.q_proj.bias is not None
        if kv_in is None:  # q, k, v from the same rows: one [3dq, dq] GEMM
            w = pc.get_cat([self.q_proj.weight, self.k_proj.weight, self.v_proj.weight], bf)
            b = pc.get_cat([self.q_proj.bias, self.k_proj.bias, self.v_proj.bias], f32) if has_b else None
            qkv = ops.gemm_bf16(q_in, w, b)
            q, k, v = qkv[:, :dq], qkv[:, dq:2 * dq], qkv[:, 2 * dq:]
        else:
            q = ops.gemm_bf16(q_in, pc.get(self.q_proj.weight, bf), pc.get(self.q_proj.bias, f32) if has_b else None)
            wkv = pc.get_cat([self.k_proj.weight, self.v_proj.weight], bf)
            bkv = pc.get_cat([self.k_proj.bias, self.v_proj.bias], f32) if has_b else None
            kv = ops.gemm_bf16(kv_in, wkv, bkv)
            k, v = kv[:, :dq], kv[:, dq:]
        present = None
        if past is not None or use_cache:
            if shared_q:
                raise ops.MmamdError("key/value caching with batch-shared queries is not a configuration of the reference")
            H = self.num_heads
            parts_k, parts_v = [], []
            if past is not None:
                pk, pv = past
                if pk.shape != pv.shape or pk.dim() != 4 or pk.shape[0] != B or pk.shape[1] != H or pk.shape[3] != hd:
                    raise ValueError(f"past_key_value must be two [bsz, num_heads, seq, head_dim] tensors, got {tuple(pk.shape)} / {tuple(pv.shape)}")
                for t, dst in ((pk, parts_k), (pv, parts_v)):
                    tok = t.detach().transpose(1, 2).reshape(B, t.shape[2], dq)  # token-major; a free view when `t` came from this module
                    if tok.dtype != bf:
                        tok = ops.convert(tok.contiguous() if tok.dtype == f32 else tok.float().contiguous(), bf)
                    dst.append(tok)
            parts_k.append(k.unflatten(0, (B, Sk)))
            parts_v.append(v.unflatten(0, (B, Sk)))
            k_all = torch.cat(parts_k, dim=1) if len(parts_k) > 1 else parts_k[0].contiguous()  # [B, Sp + Sk, dq] (data movement only)
            v_all = torch.cat(parts_v, dim=1) if len(parts_v) > 1 else parts_v[0].contiguous()
            Sk = k_all.shape[1]
            k, v = k_all.view(B * Sk, dq), v_all.view(B * Sk, dq)
            if use_cache:
                present = (k_all.view(B, Sk, H, hd).transpose(1, 2), v_all.view(B, Sk, H, hd).transpose(1, 2))
        att, _ = ops.attention_x_fwd(q, k, v, B, Sq, Sk, self.num_heads, hd, mask, shared_q=shared_q)
        y = ops.gemm_bf16(att, pc.get(self.output_proj.weight, bf), pc.get(self.output_proj.bias, f32), residual=residual,
                          out_dtype=f32, out=out)
        return (y, present) if use_cache else y

    def forward(self, query: Tensor, key: Tensor, value: Tensor, attn_mask: Optional[Tensor] = None,
                past_key_value: Optional[Tuple[Tensor, Tensor]] = None, is_causal: bool = False, use_cache: bool = False
                ) -> Union[Tensor, MHAWithCacheOutput]:
        if torch.jit.is_scripting():
            return self._forward_ops(query, key, value, attn_mask, past_key_value, is_causal, use_cache)
        else:
            return self._forward_host(query, key, value, attn_mask, past_key_value, is_causal, use_cache)

    def _run_ops(self, q_in: Tensor, k_in: Tensor, v_in: Tensor, B: int, Sq: int, Sk: int, is_causal: bool, full_mask: Optional[Tensor],
                 residual: Optional[Tensor], shared_q: bool) -> Tensor:
        """run() through the dispatcher ops (no key/value cache): q_in bf16 [B*Sq, dq] ([Sq, dq] when shared_q), k_in / v_in bf16 [B*Sk, dkv]
        -> fp32 [B*Sq, dq] = output_proj(attention) (+ residual).  Three projection GEMMs instead of the eager path's stacked one: the same
        arithmetic per output element."""
        dq = self.q_proj.out_features
        H = self.num_heads
        hd = dq // H
        if hd * H != dq or (hd != 64 and hd != 96):
            raise RuntimeError("the MI355X attention kernels are built for 64- and 96-wide heads")
        q = torch.ops.mmamd.gemm_bf16(q_in, self.q_proj.weight, self.q_proj.bias, None, 0, 1)
        k = torch.ops.mmamd.gemm_bf16(k_in, self.k_proj.weight, self.k_proj.bias, None, 0, 1)
        v = torch.ops.mmamd.gemm_bf16(v_in, self.v_proj.weight, self.v_proj.bias, None, 0, 1)
        att = torch.ops.mmamd.attn_x(q, k, v, B, Sq, Sk, H, hd, is_causal, None, full_mask, shared_q)
        return torch.ops.mmamd.gemm_bf16(att, self.output_proj.weight, self.output_proj.bias, residual, 0, 0)

    def _forward_ops(self, query: Tensor, key: Tensor, value: Tensor, attn_mask: Optional[Tensor], past_key_value: Optional[Tuple[Tensor, Tensor]],
                     is_causal: bool, use_cache: bool) -> Tensor:
        """The forward through the dispatcher ops — what torch.jit.script sees.  Boolean [Sq,Sk] / [B,Sq,Sk] masks (True = attend) or
        is_causal; incremental decoding (past_key_value / use_cache) needs the eager forward."""
        if past_key_value is not None or use_cache:
            raise RuntimeError("scripted MultiHeadAttentionWithCache on the MI355X path has no key/value cache (use the eager forward)")
        if query.dim() != 3 or key.dim() != 3 or value.dim() != 3 or key.size(0) != query.size(0) or key.size(1) != value.size(1):
            raise RuntimeError("MultiHeadAttentionWithCache takes bsz x seq_len x dim tensors with a common bsz")
        B, Sq, dq = query.size(0), query.size(1), query.size(2)
        Sk = key.size(1)
        full: Optional[Tensor] = None
        if attn_mask is not None:
            if is_causal:
                raise RuntimeError("attn_mask must be None when is_causal=True")
            if attn_mask.dtype != torch.bool or attn_mask.numel() not in (Sq * Sk, B * Sq * Sk):
                raise RuntimeError("attention masks on the MI355X path are boolean [Sq,Sk] / [B,Sq,Sk] (True = attend)")
            full = attn_mask.contiguous().to(torch.uint8)  # mask plumbing
        q_in = torch.ops.mmamd.convert(query.contiguous().view(B * Sq, dq), 1)
        k_in = torch.ops.mmamd.convert(key.contiguous().view(B * Sk, key.size(2)), 1)
        v_in = torch.ops.mmamd.convert(value.contiguous().view(B * Sk, value.size(2)), 1)
        return self._run_ops(q_in, k_in, v_in, B, Sq, Sk, is_causal, full, None, False).view(B, Sq, dq)

    @torch.jit.unused
    def _forward_host(self, query: Tensor, key: Tensor, value: Tensor, attn_mask: Optional[Tensor] = None,
                      past_key_value: Optional[Tuple[Tensor, Tensor]] = None, is_causal: bool = False, use_cache: bool = False
                      ) -> Union[Tensor, MHAWithCacheOutput]:
        if key is not value:
            raise ops.MmamdError("key and value must be the same tensor on the MI355X path (self- or cross-attention)")
        if key.size(0) != query.size(0):
            raise ValueError("key and value should have the same bsz as query.")
        forbid_detached_forward(self, query, key)
        B, Sq, dq = query.shape
        Sk = key.shape[1]
        bf = torch.bfloat16
        qc = query if query.is_contiguous() else query.contiguous()
        q_in = ops.convert(qc.view(B * Sq, dq), bf)
        kv_in = None
        if key is not query:
            kc = key if key.is_contiguous() else key.contiguous()
            kv_in = ops.convert(kc.view(B * Sk, kc.shape[-1]), bf)
        Sp = past_key_value[0].shape[2] if past_key_value is not None else 0
        mask = cached_attn_mask(attn_mask, is_causal, B, Sq, Sp + Sk, query.device)
        r = self.run(q_in, kv_in, B, Sq, Sk, mask, None, past=past_key_value, use_cache=use_cache)
        if use_cache:
            return MHAWithCacheOutput(r[0].view(B, Sq, dq), r[1])
        return r.view(B, Sq, dq)


def cached_attn_mask(attn_mask: Optional[Tensor], is_causal: bool, B: int, Sq: int, Sk: int, device) -> AttnMask:
    """to_attn_mask for a key axis that may include cached positions.  is_causal with Sq != Sk follows
    F.scaled_dot_product_attention, which the reference calls (:165-167): a TOP-LEFT aligned lower-triangular mask (query i sees keys
    0..i) — callers that decode incrementally pass an explicit mask or none, as the reference's own do."""
    if is_causal and attn_mask is None and Sq != Sk:
        attn_mask = torch.ones(Sq, Sk, dtype=torch.bool, device=device).tril()  # mask construction, not arithmetic
        is_causal = False
    return to_attn_mask(attn_mask, is_causal, B, Sq, Sk)
