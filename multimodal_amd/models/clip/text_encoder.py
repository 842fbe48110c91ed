"""CLIPTextEncoder — host-side mirror of torchmultimodal/models/clip/text_encoder.py:15-134 on the MI355X kernels.

Same constructor signature and defaults, same CLIP initialisation (text_encoder.py:82-104), same parameter
names/shapes, same ValueError on a wrong context length; forward() = token gather + pos -> causal transformer
stack -> ln_final -> EOT (argmax id) row -> projection, every step a libmmamd.so kernel.  The token-major
[B*S, w] layout replaces the reference's seq-first permutes (text_encoder.py:120,124): the math per (b, head)
is identical.
"""
from __future__ import annotations

import torch
from torch import nn, Tensor

from ... import ops
from ..._packing import PackedCache, PackedModeMixin
from ...modules.layers.normalizations import Fp32LayerNorm
from . import _train
from ._transformer import TransformerStack, forbid_training_forward


class CLIPTextEncoder(PackedModeMixin, nn.Module):
    """CLIP text encoder class. Should be instantiated and passed to CLIP (models/clip/model.py)

    Args:
        embedding_dim (int): Embedding dimension for text and image encoders projections.
        context_length (int): Maximum sequence length for Transformer.
        vocab_size (int): Vocab size.
        width (int): Embedding dimension for Transformer encoder.
        dim_feedforward (int): Dimension of the feedfoward networks.
        heads (int): Number of heads in Transformer encoder.
        layers (int): Number of layers in Transformer encoder.
        use_clip_init (bool): Whether to use CLIP-specific initialization.

    Inputs:
        text (Tensor): Tensor containing text features.
        return_hidden_state (bool): If ``True``, returns the last hidden state
            instead of the final projected embeddings. Defaults to ``False``.
    """

    TOKEN_EMBEDDING_INIT_STD = 0.02
    POS_EMBEDDING_INIT_STD = 0.01

    def __init__(self, embedding_dim: int = 512, context_length: int = 77, vocab_size: int = 49408, width: int = 512,
                 dim_feedforward: int = 2048, heads: int = 8, layers: int = 12, use_clip_init: bool = True):
        super().__init__()
        torch._C._log_api_usage_once(f"torchmultimodal.{self.__class__.__name__}")
        self.token_embedding = torch.nn.Embedding(vocab_size, width)
        self.positional_embedding = torch.nn.Parameter(torch.empty(context_length, width))
        self.encoder = TransformerStack(d_model=width, nhead=heads, dim_feedforward=dim_feedforward, num_layers=layers)
        self.width = width
        self.context_length = context_length
        self.ln_final = Fp32LayerNorm(width)
        self.projection = nn.Linear(width, embedding_dim, bias=False)
        # kept for API parity (reference: plain tensor attribute, text_encoder.py:74-77); the causal structure is
        # built into the attention kernel (upper-triangle key tiles are skipped, the diagonal tile is masked)
        self.mask = torch.full((self.context_length, self.context_length), float("-inf")).triu(1)
        if use_clip_init:
            self.initialize_parameters()
        self._packed = PackedCache()

    def initialize_parameters(self) -> None:
        nn.init.normal_(self.token_embedding.weight, std=self.TOKEN_EMBEDDING_INIT_STD)
        nn.init.normal_(self.positional_embedding, std=self.POS_EMBEDDING_INIT_STD)
        proj_std = (self.width**-0.5) * ((2 * self.encoder.num_layers) ** -0.5)
        attn_std = self.width**-0.5
        fc_std = (2 * self.width) ** -0.5
        for layer in self.encoder.layers:
            nn.init.normal_(layer.self_attn.in_proj_weight, std=attn_std)
            nn.init.normal_(layer.self_attn.out_proj.weight, std=proj_std)
            nn.init.normal_(layer.linear1.weight, std=fc_std)
            nn.init.normal_(layer.linear2.weight, std=proj_std)
        nn.init.normal_(self.projection.weight, std=self.width**-0.5)

    def build_attention_mask(self) -> Tensor:
        return torch.full((self.context_length, self.context_length), float("-inf")).triu(1)

    def forward(self, text: Tensor, return_hidden_state: bool = False) -> Tensor:
        if text.size(1) != self.context_length:
            raise ValueError(f"length of input should be {self.context_length} but found {text.size(1)}")
        if torch.jit.is_scripting():
            return self._forward_ops(text, return_hidden_state)
        else:
            return self._forward_host(text, return_hidden_state)

    def _forward_ops(self, text: Tensor, return_hidden_state: bool) -> Tensor:
        """The forward through the dispatcher ops (torch.ops.mmamd.*, csrc/torch_ops.cpp): what torch.jit.script and torch.compile see
        (reference: tests/models/clip/test_text_encoder.py:162-174 scripts this module).  Inference only."""
        B, S = text.size(0), text.size(1)
        ids = text.to(torch.int64).contiguous()
        h = torch.ops.mmamd.embed_tokens(ids, self.token_embedding.weight, self.positional_embedding)
        h = self.encoder(h, B, S, True)
        if return_hidden_state:
            hs = torch.ops.mmamd.layernorm(h, self.ln_final.weight, self.ln_final.bias, self.ln_final.eps, 0)
            return hs.view(B, S, self.width)
        return torch.ops.mmamd.pool_proj_normalize(h, B, S, ids, self.ln_final.weight, self.ln_final.bias, self.ln_final.eps,
                                                   self.projection.weight, True, False)

    @torch.jit.unused
    def _forward_host(self, text: Tensor, return_hidden_state: bool = False) -> Tensor:
        if torch.compiler.is_compiling() and not _train.wants_grad(self):
            return self._forward_ops(text, return_hidden_state)
        f32 = torch.float32
        pk = self._packed.get
        B, S = text.shape
        ids = text if (text.dtype == torch.int64 and text.is_contiguous()) else text.to(torch.int64).contiguous()
        if _train.wants_grad(self):
            h = _train.run_stack(self.encoder, self._train_stem(ids), B, S, True)
            return self._train_head(h, B, S, ids, return_hidden_state)
        h = self._stem(ids)
        h = self.encoder.run(h, B, S, causal=True)
        return self._head(h, B, S, ids, return_hidden_state)

    @torch.jit.unused
    def _train_stem(self, ids: Tensor) -> Tensor:
        """Differentiable token + positional embedding -> fp32 residual stream [B*S, w] (ids: contiguous int64 [B, S])."""
        return _train.TextEmbedFn.apply(ids, self.token_embedding.weight, self.positional_embedding)

    @torch.jit.unused
    def _train_head(self, h: Tensor, B: int, S: int, ids: Tensor, return_hidden_state: bool = False) -> Tensor:
        """Differentiable ln_final (+ projection of the end-of-text rows)."""
        if return_hidden_state:  # reference :125-127: ln_final over every token, [B, 77, width], attached to the graph (r05)
            from ..._autograd import LayerNormFn

            return LayerNormFn.apply(h, self.ln_final.weight, self.ln_final.bias, self.ln_final.eps).view(B, S, self.width)
        eot_rows = torch.arange(0, B * S, S, dtype=torch.int64, device=ids.device) + ids.argmax(dim=-1)  # index bookkeeping
        return _train.PooledHeadFn.apply(h, eot_rows, self.ln_final.weight, self.ln_final.bias, self.projection.weight,
                                         self.ln_final.eps, True)

    @torch.jit.unused
    def _stem(self, ids: Tensor) -> Tensor:
        """K8: gather + positional embedding -> fp32 residual stream [B*S, w] (ids: contiguous int64 [B, S])."""
        table = self.token_embedding.weight.detach()
        if table.dtype not in (torch.float32, torch.bfloat16):
            raise ops.MmamdError(f"token_embedding dtype {table.dtype} unsupported")
        return ops.embed_tokens(ids, table.contiguous(), self._packed.get(self.positional_embedding, torch.float32))

    @torch.jit.unused
    def _head(self, h: Tensor, B: int, S: int, ids: Tensor, return_hidden_state: bool = False) -> Tensor:
        f32 = torch.float32
        pk = self._packed.get
        out_dtype = self.projection.weight.dtype
        if return_hidden_state:
            hs = ops.layernorm(h, pk(self.ln_final.weight, f32), pk(self.ln_final.bias, f32), self.ln_final.eps,
                               out_dtype=out_dtype if out_dtype in (f32, torch.bfloat16) else f32)
            return hs.view(B, S, self.width)
        # K9: EOT row (argmax id), ln_final on that row only (identical to LN-then-gather), projection
        out = ops.pool_ln_proj(h, B, S, ids, pk(self.ln_final.weight, f32), pk(self.ln_final.bias, f32),
                               self.ln_final.eps, pk(self.projection.weight, f32), proj_is_linear_weight=True)
        return out if out_dtype == f32 else ops.convert(out, out_dtype)
