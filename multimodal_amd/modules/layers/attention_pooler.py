"""Host-side mirror of torchmultimodal/modules/layers/attention_pooler.py:16-101 (AttentionPooler, CascadedAttentionPooler).

The learned queries are the same for every sample: LayerNorm and the q projection run ONCE on [n_queries, d] and the attention
kernel reads them with batch stride 0 (the reference materialises `query.repeat(batch, 1, 1)` and projects B copies)."""

from typing import List

import torch
from torch import nn, Tensor

from ... import _torch_ops, ops
from ..._packing import PackedCache

_torch_ops.try_load()
from .multi_head_attention import MultiHeadAttentionWithCache


class AttentionPooler(nn.Module):
    def __init__(self, input_embed_dim: int, output_embed_dim: int, n_head: int, n_queries: int = 256, layer_norm_eps: float = 1e-5):
        super().__init__()
        self.query = nn.Parameter(torch.randn(n_queries, output_embed_dim))
        self.attn = MultiHeadAttentionWithCache(dim_q=output_embed_dim, dim_kv=input_embed_dim, num_heads=n_head)
        self.ln_q = nn.LayerNorm(output_embed_dim, layer_norm_eps)
        self.ln_k = nn.LayerNorm(input_embed_dim, layer_norm_eps)
        self.ln_post = nn.LayerNorm(output_embed_dim, layer_norm_eps)
        self._ln_eps: float = layer_norm_eps  # (nn.LayerNorm.eps is a TorchScript constant of the LayerNorm, not readable from here)
        self._packed = PackedCache()

    def forward(self, x: Tensor) -> Tensor:
        if torch.jit.is_scripting():
            return self._forward_ops(x)
        else:
            return self._forward_host(x)

    def _forward_ops(self, x: Tensor) -> Tensor:
        """The forward through the dispatcher ops — what torch.jit.script / torch.compile see (inference)."""
        if x.dim() != 3 or x.dtype != torch.float32:
            raise RuntimeError("AttentionPooler on the MI355X path takes fp32 [batch, seq_len, input_embed_dim] tensors")
        B, S, din = x.size(0), x.size(1), x.size(2)
        nq, dout = self.query.size(0), self.query.size(1)
        k = torch.ops.mmamd.layernorm(x.contiguous().view(B * S, din), self.ln_k.weight, self.ln_k.bias, self._ln_eps, 1)
        q = torch.ops.mmamd.layernorm(torch.ops.mmamd.packed(self.query, 0), self.ln_q.weight, self.ln_q.bias, self._ln_eps, 1)
        out = self.attn._run_ops(q, k, k, B, nq, S, False, None, None, True)
        out = torch.ops.mmamd.layernorm(out, self.ln_post.weight, self.ln_post.bias, self._ln_eps, 0)
        return out.view(B, nq, dout)

    @torch.jit.unused
    def _forward_host(self, x: Tensor) -> Tensor:
        if x.dim() != 3 or x.dtype != torch.float32:
            raise ops.MmamdError("AttentionPooler on the MI355X path takes fp32 [batch, seq_len, input_embed_dim] tensors")
        if torch.compiler.is_compiling() and not (torch.is_grad_enabled() and (x.requires_grad or (self.training and self.query.requires_grad))):
            return self._forward_ops(x)
        B, S, din = x.shape
        if torch.is_grad_enabled() and (x.requires_grad or (self.training and self.query.requires_grad)):
            # differentiable path: LayerNorm / cross-attention nodes with HIP forward and backward (queries shared by the batch)
            from ..._autograd import CrossAttentionFn, LayerNormFn

            nq, dout = self.query.shape
            k = LayerNormFn.apply(x, self.ln_k.weight, self.ln_k.bias, self.ln_k.eps).reshape(B * S, din)
            q = LayerNormFn.apply(self.query, self.ln_q.weight, self.ln_q.bias, self.ln_q.eps)
            a = self.attn
            out = CrossAttentionFn.apply(q, k, B, nq, S, a.num_heads, True, a.q_proj.weight, a.q_proj.bias, a.k_proj.weight, a.k_proj.bias,
                                         a.v_proj.weight, a.v_proj.bias, a.output_proj.weight, a.output_proj.bias)
            return LayerNormFn.apply(out.view(B, nq, dout), self.ln_post.weight, self.ln_post.bias, self.ln_post.eps)
        pk, bf, f32 = self._packed.get, torch.bfloat16, torch.float32
        xc = (x if x.is_contiguous() else x.contiguous()).view(B * S, din)
        k = ops.layernorm(xc, pk(self.ln_k.weight, f32), pk(self.ln_k.bias, f32), self.ln_k.eps, out_dtype=bf)
        q = ops.layernorm(pk(self.query, f32), pk(self.ln_q.weight, f32), pk(self.ln_q.bias, f32), self.ln_q.eps, out_dtype=bf)
        nq, dout = self.query.shape
        out = self.attn.run(q, k, B, nq, S, ops.AttnMask(), None, shared_q=True)
        out = ops.layernorm(out, pk(self.ln_post.weight, f32), pk(self.ln_post.bias, f32), self.ln_post.eps, out_dtype=f32)
        return out.view(B, nq, dout)

    def _repeat(self, query: Tensor, n: int) -> Tensor:
        return query.unsqueeze(0).repeat(n, 1, 1)


class CascadedAttentionPooler(nn.Module):
    """Cascaded pooling: each pooler runs on the previous pooler's output (CoCa: captioning pooler, then contrastive pooler)."""

    def __init__(self, poolers: List[AttentionPooler]):
        super().__init__()
        self.poolers = nn.ModuleList(poolers)

    def forward(self, x: Tensor) -> List[Tensor]:
        pooler_outs = []
        for pooler in self.poolers:
            x = pooler(x)
            pooler_outs.append(x)
        return pooler_outs
