"""Host-side mirror of torchmultimodal/models/coca/multimodal_decoder.py:14-108 (CoCaMultimodalDecoder): causal transformer
decoder over the text tokens that cross-attends to the captioning image embeddings, then the (vocabulary) output projection —
one bf16 MFMA GEMM with fp32 logits."""

from typing import Callable, Optional

import torch
from torch import nn, Tensor

from ... import _torch_ops, ops
from ..._packing import PackedCache
from ...modules.layers.transformer import TransformerDecoder
from ...utils.attention import get_causal_attention_mask


_torch_ops.try_load()


class CoCaMultimodalDecoder(nn.Module):
    def __init__(self, input_seq_len: int, text_embedding_dim: int, n_layer: int, n_head: int, dim_feedforward: int,
                 output_dim: Optional[int] = None, dropout: float = 0.0, activation: Callable[..., nn.Module] = nn.GELU,
                 layer_norm_eps: float = 1e-5, norm_first: bool = True, final_layer_norm_eps: Optional[float] = 1e-5,
                 visual_embedding_dim: Optional[int] = None):
        super().__init__()
        self.transformer_decoder = TransformerDecoder(
            n_layer=n_layer, d_model=text_embedding_dim, n_head=n_head, dim_feedforward=dim_feedforward, dropout=dropout,
            activation=activation, layer_norm_eps=layer_norm_eps, norm_first=norm_first, use_cross_attention=True,
            final_layer_norm_eps=final_layer_norm_eps, dim_kv=visual_embedding_dim)
        if output_dim is not None:
            self.output_projection = nn.Linear(text_embedding_dim, output_dim, bias=False)
        else:
            self.output_projection = None
        self.register_buffer("causal_mask", get_causal_attention_mask(input_seq_len).to(dtype=torch.bool), persistent=False)
        self._packed = PackedCache()

    def forward(self, texts: Tensor, images: Tensor) -> Tensor:
        if torch.jit.is_scripting():
            return self._forward_ops(texts, images)
        else:
            return self._forward_host(texts, images)

    def _forward_ops(self, texts: Tensor, images: Tensor) -> Tensor:
        """The forward through the dispatcher ops — what torch.jit.script / torch.compile see (inference)."""
        seq_len = texts.size(1)
        assert self.causal_mask.size(0) == seq_len and self.causal_mask.size(1) == seq_len
        hidden_states = self.transformer_decoder._forward_ops(texts, images, True, None, False).last_hidden_state
        assert hidden_states is not None, "hidden states must not be None"
        proj = self.output_projection
        if proj is None:
            return hidden_states
        B, S, d = hidden_states.size(0), hidden_states.size(1), hidden_states.size(2)
        V = proj.weight.size(0)
        if V % 8 != 0:
            raise RuntimeError("scripted CoCaMultimodalDecoder on the MI355X path: output_dim must be a multiple of 8 (the eager forward pads it)")
        h = torch.ops.mmamd.convert(hidden_states.contiguous().view(B * S, d), 1)
        return torch.ops.mmamd.gemm_bf16(h, proj.weight, None, None, 0, 0).view(B, S, V)

    @torch.jit.unused
    def _forward_host(self, texts: Tensor, images: Tensor) -> Tensor:
        if torch.compiler.is_compiling() and not torch.is_grad_enabled() and (self.output_projection is None or self.output_projection.out_features % 8 == 0):
            return self._forward_ops(texts, images)
        seq_len = texts.shape[1]
        assert self.causal_mask.shape == (seq_len, seq_len)
        # the registered causal_mask buffer IS lower-triangular: the kernel's causal flag skips the key tiles above the diagonal
        decoder_outputs = self.transformer_decoder(hidden_states=texts, encoder_hidden_states=images,
                                                   attention_mask=ops.AttnMask(causal=True))
        hidden_states = decoder_outputs.last_hidden_state
        assert hidden_states is not None, "hidden states must not be None"
        if self.output_projection is None:
            return hidden_states
        B, S, d = hidden_states.shape
        if torch.is_grad_enabled() and hidden_states.requires_grad:
            from ..._autograd import LinearFn  # differentiable vocabulary projection (dgrad + split-K wgrad)

            return LinearFn.apply(hidden_states.reshape(B * S, d), self.output_projection.weight, None).unflatten(0, (B, S))
        h = ops.convert(hidden_states.view(B * S, d), torch.bfloat16)
        V = self.output_projection.out_features
        w = self._packed.get_padded_rows(self.output_projection.weight, torch.bfloat16, 8)
        out = ops.gemm_bf16(h, w, None, out_dtype=torch.float32)
        out = out if out.shape[1] == V else out[:, :V]
        return out.unflatten(0, (B, S))
