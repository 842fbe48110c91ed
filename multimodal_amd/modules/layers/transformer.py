"""Host-side mirror of torchmultimodal/modules/layers/transformer.py: the output record every encoder returns (:21-27) and the
TransformerEncoder(Layer) / TransformerDecoder(Layer) blocks (:31-657) CoCa's ViT (encoders/vision_transformer.py), text decoder and
multimodal decoder are built from.  Same constructors, attribute names and state_dict keys.  Per layer on the MI355X (pre-norm):

    LN -> [3d,d] GEMM -> attention -> output GEMM(+residual)   [-> LN -> q GEMM, [2d,dkv] GEMM -> cross-attention -> GEMM(+res)]
    LN -> up GEMM(+bias, GELU) -> down GEMM(+bias, +residual)

residual stream fp32, MFMA operands bf16.
"""
from typing import Callable, List, NamedTuple, Optional, Tuple

import torch
from torch import nn, Tensor

from ... import _torch_ops, ops
from ..._autograd import params_require_grad, plain_layers, wants_grad
from ..._packing import PackedCache
from .mlp import MLP
from .multi_head_attention import MultiHeadAttentionWithCache, MultiHeadSelfAttention, to_attn_mask
from .normalizations import Fp32LayerNorm
from .stochastic_depth import StochasticDepth


_torch_ops.try_load()


class TransformerOutput(NamedTuple):
    last_hidden_state: Optional[Tensor] = None
    pooler_output: Optional[Tensor] = None
    hidden_states: Optional[List[Tensor]] = None
    attentions: Optional[List[Tensor]] = None
    image_labels: Optional[Tensor] = None
    current_key_values: Optional[List[Tuple[Tensor, Tensor]]] = None


def _forbid_training(module: nn.Module) -> None:
    if module.training and torch.is_grad_enabled() and params_require_grad(module):
        raise NotImplementedError(f"{type(module).__name__}: a standalone layer has no differentiable forward on the MI355X path (training runs "
                                  "through TransformerEncoder / TransformerDecoder); call .eval() and/or run under torch.no_grad()")


def _ln(packed: PackedCache, ln: nn.LayerNorm, x: Tensor, out_dtype: torch.dtype) -> Tensor:
    return ops.layernorm(x, packed.get(ln.weight, torch.float32), packed.get(ln.bias, torch.float32), ln.eps, out_dtype=out_dtype)


def _f32_rows(t: Tensor, what: str) -> Tensor:
    if t.dim() != 3 or t.dtype != torch.float32:
        raise ops.MmamdError(f"{what} on the MI355X path takes fp32 [bsz, seq_len, d_model] tensors")
    return (t if t.is_contiguous() else t.contiguous()).view(-1, t.shape[-1])


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model: int, n_head: int, dim_feedforward: int, dropout: float = 0.0,
                 activation: Callable[..., nn.Module] = nn.ReLU, layer_norm_eps: float = 1e-12, norm_first: bool = False,
                 drop_path_rate: Optional[float] = None) -> None:
        super().__init__()
        self.attention = MultiHeadSelfAttention(embed_dim=d_model, num_heads=n_head)
        if drop_path_rate is not None:  # reference :64-67: ONE StochasticDepth(mode="row") serves both residual branches
            self.attention_dropout = self.feedforward_dropout = StochasticDepth(drop_path_rate, mode="row")
        else:
            self.attention_dropout = nn.Dropout(dropout)
            self.feedforward_dropout = nn.Dropout(dropout)
        self.feedforward = MLP(d_model, d_model, dim_feedforward, dropout=dropout, activation=activation)
        self.attention_layernorm = Fp32LayerNorm(d_model, eps=layer_norm_eps)
        self.feedforward_layernorm = Fp32LayerNorm(d_model, eps=layer_norm_eps)
        self.norm_first = norm_first
        self._packed = PackedCache()

    def run(self, x: Tensor, B: int, S: int, mask: ops.AttnMask) -> Tensor:
        """x: fp32 [B*S, d] (left untouched) -> new fp32 [B*S, d].  INFERENCE form (no dropout / stochastic depth): the training forward of the
        enclosing TransformerEncoder applies them (_encoder_forward_train); a train-mode module with non-zero rates must not land here."""
        if self.training and (getattr(self.attention_dropout, "p", 0.0) > 0 or getattr(self.feedforward_dropout, "p", 0.0) > 0
                              or self.feedforward.hidden_dropout_p() > 0):
            raise ops.MmamdError("this non-differentiable forward applies no dropout / stochastic depth: call .eval() for inference (training goes "
                                 "through TransformerEncoder's differentiable forward, which does apply them)")
        bf, f32, pc = torch.bfloat16, torch.float32, self._packed
        if self.norm_first:  # reference :96-116
            x1 = self.attention.run(_ln(pc, self.attention_layernorm, x, bf), B, S, mask, residual=x)
            return self.feedforward.run(_ln(pc, self.feedforward_layernorm, x1, bf), residual=x1, out=x1)
        a = self.attention.run(ops.convert(x, bf), B, S, mask, residual=x)  # post-norm, :118-132
        x1 = _ln(pc, self.attention_layernorm, a, f32)
        ff = self.feedforward.run(ops.convert(x1, bf), residual=x1, out=a)
        return _ln(pc, self.feedforward_layernorm, ff, f32)

    def forward(self, hidden_states: Tensor, attention_mask: Optional[Tensor] = None) -> Tensor:
        if torch.jit.is_scripting():
            return self._forward_ops(hidden_states, attention_mask)
        else:
            return self._forward_host(hidden_states, attention_mask)

    def _layer_ops(self, x: Tensor, B: int, S: int) -> Tensor:
        """run() through the dispatcher ops (torch.ops.mmamd.*; pre-norm layers, no mask): x fp32 [B*S, d] -> new fp32 [B*S, d]."""
        if not self.norm_first:
            raise RuntimeError("scripted TransformerEncoderLayer on the MI355X path: pre-norm layers only (post-norm uses the eager forward)")
        hn = self.attention_layernorm._ln_ops(x, 1)
        x1 = self.attention._run_ops(hn, B, S, False, x)
        return self.feedforward._run_ops(self.feedforward_layernorm._ln_ops(x1, 1), x1)

    def _forward_ops(self, hidden_states: Tensor, attention_mask: Optional[Tensor]) -> Tensor:
        if attention_mask is not None:
            raise RuntimeError("scripted TransformerEncoderLayer on the MI355X path takes no attention_mask (use the eager forward)")
        if hidden_states.dim() != 3 or hidden_states.dtype != torch.float32:
            raise RuntimeError("TransformerEncoderLayer on the MI355X path takes fp32 [bsz, seq_len, d_model] tensors")
        B, S, d = hidden_states.size(0), hidden_states.size(1), hidden_states.size(2)
        return self._layer_ops(hidden_states.contiguous().view(B * S, d), B, S).view(B, S, d)

    @torch.jit.unused
    def _forward_host(self, hidden_states: Tensor, attention_mask: Optional[Tensor] = None) -> Tensor:
        if wants_grad(self) or (torch.is_grad_enabled() and hidden_states.requires_grad):
            # differentiable stand-alone layer (a one-layer EncoderStackFn node): what a wrapped layer (FSDP, checkpoint_wrapper) or a
            # user's own stack of layers runs in training
            return _layers_forward_train([self], self.training, None, hidden_states, attention_mask, False).last_hidden_state
        B, S, d = hidden_states.shape
        y = self.run(_f32_rows(hidden_states, "TransformerEncoderLayer"), B, S, to_attn_mask(attention_mask, False, B, S, S))
        return y.view(B, S, d)


class TransformerEncoder(nn.Module):
    def __init__(self, n_layer: int, d_model: int, n_head: int, dim_feedforward: int, dropout: float = 0.0,
                 activation: Callable[..., nn.Module] = nn.ReLU, layer_norm_eps: float = 1e-12, norm_first: bool = False,
                 final_layer_norm_eps: Optional[float] = None, drop_path_rate: Optional[float] = None):
        super().__init__()
        if drop_path_rate is not None:  # reference :190-193: the rate grows linearly with depth
            drop_rate = [x.item() for x in torch.linspace(0, drop_path_rate, n_layer)]
        else:
            drop_rate = [None for _ in range(n_layer)]
        self.layer = nn.ModuleList([
            TransformerEncoderLayer(d_model, n_head, dim_feedforward, dropout, activation, layer_norm_eps, norm_first, drop_rate[i])
            for i in range(n_layer)
        ])
        self.final_layer_norm = None
        if final_layer_norm_eps:
            self.final_layer_norm = Fp32LayerNorm(d_model, eps=final_layer_norm_eps)

    def forward(self, hidden_states: Tensor, attention_mask: Optional[Tensor] = None, return_hidden_states: bool = False
                ) -> TransformerOutput:
        if torch.jit.is_scripting():
            return self._forward_ops(hidden_states, attention_mask, return_hidden_states)
        else:
            return self._forward_host(hidden_states, attention_mask, return_hidden_states)

    def _forward_ops(self, hidden_states: Tensor, attention_mask: Optional[Tensor], return_hidden_states: bool) -> TransformerOutput:
        """The forward through the dispatcher ops — what torch.jit.script / torch.compile see (inference, pre-norm, no mask)."""
        if attention_mask is not None:
            raise RuntimeError("scripted TransformerEncoder on the MI355X path takes no attention_mask (use the eager forward)")
        if hidden_states.dim() != 3 or hidden_states.dtype != torch.float32:
            raise RuntimeError("TransformerEncoder on the MI355X path takes fp32 [bsz, seq_len, d_model] tensors")
        B, S, d = hidden_states.size(0), hidden_states.size(1), hidden_states.size(2)
        x = hidden_states.contiguous().view(B * S, d)
        all_hidden_states: List[Tensor] = []
        for layer_module in self.layer:
            if return_hidden_states:
                all_hidden_states.append(x.view(B, S, d))
            x = layer_module._layer_ops(x, B, S)
        y = x.view(B, S, d)
        hs: Optional[List[Tensor]] = None
        if return_hidden_states:
            all_hidden_states.append(y)
            hs = all_hidden_states
        if self.final_layer_norm is not None:
            y = self.final_layer_norm(y)
        return TransformerOutput(last_hidden_state=y, hidden_states=hs)

    @torch.jit.unused
    def _forward_host(self, hidden_states: Tensor, attention_mask: Optional[Tensor] = None, return_hidden_states: bool = False
                      ) -> TransformerOutput:
        B, S, d = hidden_states.shape
        if not plain_layers(self.layer, TransformerEncoderLayer):
            return self._forward_by_module(hidden_states, attention_mask, return_hidden_states)
        if wants_grad(self) or (torch.is_grad_enabled() and hidden_states.requires_grad):
            return self._forward_train(hidden_states, attention_mask, return_hidden_states)
        if torch.compiler.is_compiling():
            return self._forward_ops(hidden_states, attention_mask, return_hidden_states)
        x = _f32_rows(hidden_states, "TransformerEncoder")
        mask = to_attn_mask(attention_mask, False, B, S, S)
        all_hidden_states = []
        for layer_module in self.layer:
            if return_hidden_states:
                all_hidden_states.append(x.view(B, S, d))
            x = layer_module.run(x, B, S, mask)
        x = x.view(B, S, d)
        if return_hidden_states:
            all_hidden_states.append(x)
        if self.final_layer_norm is not None:
            x = self.final_layer_norm(x)
        return TransformerOutput(last_hidden_state=x, hidden_states=all_hidden_states if return_hidden_states else None)


def _encoder_forward_by_module(self, hidden_states: Tensor, attention_mask, return_hidden_states: bool) -> TransformerOutput:
    """The reference's own loop (transformer.py:230-247): every layer CALLED as a module, so that wrappers (FSDP unshards a layer's
    parameters around its forward; checkpoint_wrapper) and hooks see the call.  Each layer decides between its inference kernels and its
    one-layer autograd node on its own; hidden states are the layers' outputs, attached to the graph in training."""
    x = hidden_states
    all_hidden_states = []
    for layer_module in self.layer:
        if return_hidden_states:
            all_hidden_states.append(x)
        x = layer_module(x, attention_mask=attention_mask)
    if return_hidden_states:
        all_hidden_states.append(x)
    if self.final_layer_norm is not None:
        x = self.final_layer_norm(x)
    return TransformerOutput(last_hidden_state=x, hidden_states=all_hidden_states if return_hidden_states else None)


TransformerEncoder._forward_by_module = _encoder_forward_by_module


def _encoder_forward_train(self, hidden_states: Tensor, attention_mask, return_hidden_states: bool) -> TransformerOutput:
    """Differentiable TransformerEncoder.forward: the packed input_proj layout is the canonical one of EncoderStackFn."""
    return _layers_forward_train(list(self.layer), self.training, self.final_layer_norm, hidden_states, attention_mask, return_hidden_states)


def _layers_forward_train(layers, training: bool, final_layer_norm, hidden_states: Tensor, attention_mask,
                          return_hidden_states: bool) -> TransformerOutput:
    """Differentiable forward of a list of TransformerEncoderLayers (a whole TransformerEncoder, or ONE stand-alone layer) as one
    EncoderStackFn node.  Pre-norm and post-norm (the reference's default, transformer.py:56) layers; attention_mask: None, or any mask that
    the reference's boolean masks (to_attn_mask: [S, S], [B, S, S], [B, 1, S, S]; they go through the general attention kernels)."""
    from ..._autograd import EncoderStackFn, StackConfig, stack_drop_spec
    from .mlp import fused_activation_code  # noqa: F401

    B, S, d = hidden_states.shape
    if hidden_states.dtype != torch.float32:
        raise ops.MmamdError("TransformerEncoder on the MI355X path takes fp32 [bsz, seq_len, d_model] tensors")
    mask = to_attn_mask(attention_mask, False, B, S, S)
    norm_first = {bool(layer.norm_first) for layer in layers}
    if len(norm_first) != 1:
        raise ops.MmamdError("training: all layers of a stack must share norm_first")
    params, eps1, eps2, act = [], [], [], None
    for layer in layers:
        steps = layer.feedforward.plan()
        if len(steps) != 2 or steps[1][1] != ops.ACT_NONE or steps[0][1] not in (ops.ACT_GELU_ERF, ops.ACT_QUICKGELU):
            raise ops.MmamdError("training: the feed-forward block must be Linear -> GELU/QuickGELU -> Linear")
        act = steps[0][1]
        at = layer.attention
        params += [at.input_proj.weight, at.input_proj.bias, at.output_proj.weight, at.output_proj.bias, steps[0][0].weight,
                   steps[0][0].bias, steps[1][0].weight, steps[1][0].bias, layer.attention_layernorm.weight,
                   layer.attention_layernorm.bias, layer.feedforward_layernorm.weight, layer.feedforward_layernorm.bias]
        eps1.append(layer.attention_layernorm.eps)
        eps2.append(layer.feedforward_layernorm.eps)
    ident = lambda t: t
    # training-time dropout / stochastic depth of the residual branches and the MLP's hidden dropout (reference :64-93); MultiHeadSelfAttention is
    # built without attention-probability dropout (:60-63)
    drop, seed = stack_drop_spec(layers, training=training)
    cfg = StackConfig(len(layers), layers[0].attention.num_heads, B, S, bool(mask.causal), act, eps1, eps2, 12, ident, ident,
                      key_mask=mask.key_mask, keep_hidden=return_hidden_states, drop=drop, seed=seed, norm_first=norm_first.pop(),
                      full_mask=mask.full)
    xc = hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous()
    res = EncoderStackFn.apply(xc.view(B * S, d), cfg, *params)
    hidden = None
    if return_hidden_states:  # every entry attached to the graph, like the reference's (:230-247): the input, the layer inputs, the result
        x = res[0].view(B, S, d)
        hidden = [hidden_states] + [h.view(B, S, d) for h in res[1:]] + [x]
    else:
        x = res.view(B, S, d)
    if final_layer_norm is not None:
        x = final_layer_norm(x)
    return TransformerOutput(last_hidden_state=x, hidden_states=hidden)


TransformerEncoder._forward_train = _encoder_forward_train


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model: int, n_head: int, dim_feedforward: int, dropout: float = 0.0,
                 activation: Callable[..., nn.Module] = nn.ReLU, layer_norm_eps: float = 1e-12, norm_first: bool = False,
                 use_cross_attention: bool = True, dim_kv: Optional[int] = None) -> None:
        super().__init__()
        dim_kv = dim_kv if dim_kv is not None else d_model
        self.attention = MultiHeadAttentionWithCache(dim_q=d_model, dim_kv=d_model, num_heads=n_head, dropout=dropout)
        self.attention_dropout = nn.Dropout(dropout)
        self.cross_attention: Optional[MultiHeadAttentionWithCache] = None
        self.use_cross_attention = use_cross_attention
        if self.use_cross_attention:
            self.cross_attention = MultiHeadAttentionWithCache(dim_q=d_model, dim_kv=dim_kv, num_heads=n_head, dropout=dropout)
            self.cross_attention_layernorm = Fp32LayerNorm(d_model, eps=layer_norm_eps)
            self.cross_attention_dropout = nn.Dropout(dropout)
        self.feedforward = MLP(d_model, d_model, dim_feedforward, dropout=dropout, activation=activation)
        self.feedforward_dropout = nn.Dropout(dropout)
        self.attention_layernorm = Fp32LayerNorm(d_model, eps=layer_norm_eps)
        self.feedforward_layernorm = Fp32LayerNorm(d_model, eps=layer_norm_eps)
        self.norm_first = norm_first
        self._packed = PackedCache()

    def run(self, x: Tensor, B: int, S: int, mask: ops.AttnMask, enc: Optional[Tensor] = None, Sk: int = 0,
            cross_mask: Optional[ops.AttnMask] = None, past: Optional[Tuple[Tensor, Tensor]] = None, use_cache: bool = False):
        """x: fp32 [B*S, d]; enc: bf16 [B*Sk, dim_kv] encoder states (already converted once per decoder) or None.  With `past` /
        use_cache the self-attention runs over the cached + new keys (reference :336-359) and the result is (y, present)."""
        if self.training and (self.attention_dropout.p > 0 or self.feedforward_dropout.p > 0):
            raise ops.MmamdError("this non-differentiable forward applies no dropout: call .eval() for inference (training goes through "
                                 "TransformerDecoder's differentiable forward, which does apply it)")
        bf, f32, pc = torch.bfloat16, torch.float32, self._packed
        cross_mask = cross_mask or ops.AttnMask()
        caching = past is not None or use_cache
        present = None

        def self_attn(h_in, residual):
            nonlocal present
            r = self.attention.run(h_in, None, B, S, S, mask, residual=residual, past=past, use_cache=use_cache)
            if use_cache:
                r, present = r
            return r

        if self.norm_first:  # reference :398-433
            a = self_attn(_ln(pc, self.attention_layernorm, x, bf), x)
            if self.use_cross_attention and enc is not None:
                a = self.cross_attention.run(_ln(pc, self.cross_attention_layernorm, a, bf), enc, B, S, Sk, cross_mask, residual=a, out=a)
            y = self.feedforward.run(_ln(pc, self.feedforward_layernorm, a, bf), residual=a, out=a)
            return (y, present) if caching else y
        # post-norm, :435-472
        a = self_attn(ops.convert(x, bf), x)
        a = _ln(pc, self.attention_layernorm, a, f32)
        if self.use_cross_attention:
            if enc is None:
                raise ValueError("encoder_hidden_states must be provided for cross attention")
            c = self.cross_attention.run(ops.convert(a, bf), enc, B, S, Sk, cross_mask, residual=a)
            a = _ln(pc, self.cross_attention_layernorm, c, f32)
        ff = self.feedforward.run(ops.convert(a, bf), residual=a)
        y = _ln(pc, self.feedforward_layernorm, ff, f32)
        return (y, present) if caching else y

    def forward(self, hidden_states: Tensor, encoder_hidden_states: Optional[Tensor] = None, attention_mask: Optional[Tensor] = None,
                cross_attention_mask: Optional[Tensor] = None, past_key_value: Optional[Tuple[Tensor, Tensor]] = None,
                use_cache: bool = False) -> Tuple[Tensor, Optional[Tuple[Tensor, Tensor]]]:
        if torch.jit.is_scripting():
            return self._forward_ops(hidden_states, encoder_hidden_states, attention_mask, cross_attention_mask, past_key_value, use_cache)
        else:
            return self._forward_host(hidden_states, encoder_hidden_states, attention_mask, cross_attention_mask, past_key_value, use_cache)

    def _layer_ops(self, x: Tensor, B: int, S: int, is_causal: bool, full_mask: Optional[Tensor], enc: Optional[Tensor], Sk: int) -> Tensor:
        """run() through the dispatcher ops (pre-norm, no cache): x fp32 [B*S, d]; full_mask uint8 [B or 1, S, S] (0 = masked) or is_causal;
        enc bf16 [B*Sk, dim_kv] encoder states or None -> new fp32 [B*S, d]."""
        if not self.norm_first:
            raise RuntimeError("scripted TransformerDecoderLayer on the MI355X path: pre-norm layers only (post-norm uses the eager forward)")
        hn = self.attention_layernorm._ln_ops(x, 1)
        a = self.attention._run_ops(hn, hn, hn, B, S, S, is_causal, full_mask, x, False)
        if self.cross_attention is not None:
            if enc is not None:
                hc = self.cross_attention_layernorm._ln_ops(a, 1)
                a = self.cross_attention._run_ops(hc, enc, enc, B, S, Sk, False, None, a, False)
        return self.feedforward._run_ops(self.feedforward_layernorm._ln_ops(a, 1), a)

    def _forward_ops(self, hidden_states: Tensor, encoder_hidden_states: Optional[Tensor], attention_mask: Optional[Tensor],
                     cross_attention_mask: Optional[Tensor], past_key_value: Optional[Tuple[Tensor, Tensor]], use_cache: bool
                     ) -> Tuple[Tensor, Optional[Tuple[Tensor, Tensor]]]:
        if past_key_value is not None or use_cache:
            raise RuntimeError("scripted TransformerDecoderLayer on the MI355X path has no key/value cache (use the eager forward)")
        if cross_attention_mask is not None:
            raise RuntimeError("scripted TransformerDecoderLayer on the MI355X path takes no cross_attention_mask")
        if hidden_states.dim() != 3 or hidden_states.dtype != torch.float32:
            raise RuntimeError("TransformerDecoderLayer on the MI355X path takes fp32 [bsz, seq_len, d_model] tensors")
        B, S, d = hidden_states.size(0), hidden_states.size(1), hidden_states.size(2)
        enc: Optional[Tensor] = None
        Sk = 0
        if encoder_hidden_states is not None:
            Sk = encoder_hidden_states.size(1)
            enc = torch.ops.mmamd.convert(encoder_hidden_states.contiguous().view(B * Sk, encoder_hidden_states.size(2)), 1)
        full: Optional[Tensor] = None
        if attention_mask is not None:
            if attention_mask.dtype != torch.bool or attention_mask.numel() not in (S * S, B * S * S):
                raise RuntimeError("attention masks on the MI355X path are boolean [S,S] / [B,S,S] (True = attend)")
            full = attention_mask.contiguous().to(torch.uint8)  # mask plumbing
        y = self._layer_ops(hidden_states.contiguous().view(B * S, d), B, S, False, full, enc, Sk)
        none_kv: Optional[Tuple[Tensor, Tensor]] = None
        return y.view(B, S, d), none_kv

    @torch.jit.unused
    def _forward_host(self, hidden_states: Tensor, encoder_hidden_states: Optional[Tensor] = None, attention_mask: Optional[Tensor] = None,
                      cross_attention_mask: Optional[Tensor] = None, past_key_value: Optional[Tuple[Tensor, Tensor]] = None,
                      use_cache: bool = False) -> Tuple[Tensor, Optional[Tuple[Tensor, Tensor]]]:
        if wants_grad(self) or (torch.is_grad_enabled() and hidden_states.requires_grad):
            # differentiable stand-alone layer: a one-layer DecoderStackFn node (what a wrapped layer runs in training)
            if past_key_value is not None or use_cache:
                raise ops.MmamdError("key/value caching is an inference feature: call the layer under torch.no_grad() / in eval mode")
            out = _decoder_layers_forward_train([self], self.training, None, hidden_states, encoder_hidden_states, attention_mask, False,
                                                cross_attention_mask=cross_attention_mask)
            return out.last_hidden_state, None
        B, S, d = hidden_states.shape
        enc, Sk = None, 0
        if encoder_hidden_states is not None:
            Sk = encoder_hidden_states.shape[1]
            enc = ops.convert(_f32_rows(encoder_hidden_states, "TransformerDecoderLayer"), torch.bfloat16)
        St = S + (past_key_value[0].shape[2] if past_key_value is not None else 0)  # keys the self-attention mask spans
        r = self.run(_f32_rows(hidden_states, "TransformerDecoderLayer"), B, S, to_attn_mask(attention_mask, False, B, S, St), enc, Sk,
                     to_attn_mask(cross_attention_mask, False, B, S, Sk) if enc is not None else None, past=past_key_value, use_cache=use_cache)
        if past_key_value is not None or use_cache:
            return r[0].view(B, S, d), r[1]
        return r.view(B, S, d), None


class TransformerDecoder(nn.Module):
    def __init__(self, n_layer: int, d_model: int, n_head: int, dim_feedforward: int, dropout: float = 0.0,
                 activation: Callable[..., nn.Module] = nn.ReLU, layer_norm_eps: float = 1e-12, norm_first: bool = False,
                 use_cross_attention: bool = True, dim_kv: Optional[int] = None, final_layer_norm_eps: Optional[float] = None,
                 cross_attention_interval: int = 1):
        super().__init__()
        self.layer = nn.ModuleList([
            TransformerDecoderLayer(d_model, n_head, dim_feedforward, dropout, activation, layer_norm_eps, norm_first,
                                    use_cross_attention and (i % cross_attention_interval == 0), dim_kv)
            for i in range(n_layer)
        ])
        self.final_layer_norm = None
        if final_layer_norm_eps:
            self.final_layer_norm = Fp32LayerNorm(d_model, eps=final_layer_norm_eps)

    def forward(self, hidden_states: Tensor, encoder_hidden_states: Optional[Tensor] = None, attention_mask: Optional[Tensor] = None,
                cross_attention_mask: Optional[Tensor] = None, past_key_values: Optional[List[Tuple[Tensor, Tensor]]] = None,
                use_cache: bool = False, return_hidden_states: bool = False) -> TransformerOutput:
        if torch.jit.is_scripting():
            if past_key_values is not None or use_cache:
                raise RuntimeError("scripted TransformerDecoder on the MI355X path has no key/value cache (use the eager forward)")
            full: Optional[Tensor] = None
            if attention_mask is not None:
                if attention_mask.dtype != torch.bool:
                    raise RuntimeError("attention masks on the MI355X path are boolean (True = attend)")
                full = attention_mask.contiguous().to(torch.uint8)  # mask plumbing
            return self._forward_ops(hidden_states, encoder_hidden_states, False, full, return_hidden_states)
        else:
            return self._forward_host(hidden_states, encoder_hidden_states, attention_mask, cross_attention_mask, past_key_values, use_cache,
                                      return_hidden_states)

    def _forward_ops(self, hidden_states: Tensor, encoder_hidden_states: Optional[Tensor], is_causal: bool, full_mask: Optional[Tensor],
                     return_hidden_states: bool) -> TransformerOutput:
        """The forward through the dispatcher ops (inference, pre-norm, no cache): self-attention mask = is_causal or a uint8
        [B or 1, S, S] mask (0 = masked), as CoCa's decoders build them."""
        if hidden_states.dim() != 3 or hidden_states.dtype != torch.float32:
            raise RuntimeError("TransformerDecoder on the MI355X path takes fp32 [bsz, seq_len, d_model] tensors")
        B, S, d = hidden_states.size(0), hidden_states.size(1), hidden_states.size(2)
        x = hidden_states.contiguous().view(B * S, d)
        enc: Optional[Tensor] = None
        Sk = 0
        if encoder_hidden_states is not None:
            if encoder_hidden_states.dim() != 3 or encoder_hidden_states.dtype != torch.float32:
                raise RuntimeError("TransformerDecoder on the MI355X path takes fp32 [bsz, seq_len, dim_kv] encoder states")
            Sk = encoder_hidden_states.size(1)
            enc = torch.ops.mmamd.convert(encoder_hidden_states.contiguous().view(B * Sk, encoder_hidden_states.size(2)), 1)  # once for all layers
        all_hidden_states: List[Tensor] = []
        for layer_module in self.layer:
            if return_hidden_states:
                all_hidden_states.append(x.view(B, S, d))
            x = layer_module._layer_ops(x, B, S, is_causal, full_mask, enc, Sk)
        y = x.view(B, S, d)
        if return_hidden_states:
            all_hidden_states.append(y)
        if self.final_layer_norm is not None:
            y = self.final_layer_norm(y)
        kv: List[Tuple[Tensor, Tensor]] = []
        return TransformerOutput(last_hidden_state=y, hidden_states=all_hidden_states, current_key_values=kv)

    @torch.jit.unused
    def _forward_host(self, hidden_states: Tensor, encoder_hidden_states: Optional[Tensor] = None, attention_mask: Optional[Tensor] = None,
                      cross_attention_mask: Optional[Tensor] = None, past_key_values: Optional[List[Tuple[Tensor, Tensor]]] = None,
                      use_cache: bool = False, return_hidden_states: bool = False) -> TransformerOutput:
        B, S, d = hidden_states.shape
        caching = past_key_values is not None or use_cache
        if not plain_layers(self.layer, TransformerDecoderLayer):
            # the reference's own loop (:606-640), every layer CALLED as a module (wrapped / hooked layers)
            x = hidden_states
            all_hidden_states, current_key_values = [], []
            for i, layer_module in enumerate(self.layer):
                if return_hidden_states:
                    all_hidden_states.append(x)
                x, present = layer_module(x, encoder_hidden_states, attention_mask=attention_mask,
                                          past_key_value=past_key_values[i] if past_key_values is not None else None, use_cache=use_cache)
                if use_cache:
                    current_key_values.append(present)
            if return_hidden_states:
                all_hidden_states.append(x)
            if self.final_layer_norm is not None:
                x = self.final_layer_norm(x)
            return TransformerOutput(last_hidden_state=x, hidden_states=all_hidden_states, current_key_values=current_key_values)
        if wants_grad(self) or (torch.is_grad_enabled() and hidden_states.requires_grad):
            if caching:
                raise ops.MmamdError("key/value caching is an inference feature: call the decoder under torch.no_grad() / in eval mode")
            return self._forward_train(hidden_states, encoder_hidden_states, attention_mask, return_hidden_states)
        x = _f32_rows(hidden_states, "TransformerDecoder")
        if past_key_values is not None and len(past_key_values) != len(self.layer):
            raise ValueError(f"past_key_values has {len(past_key_values)} entries for {len(self.layer)} layers")
        St = S + (past_key_values[0][0].shape[2] if past_key_values is not None else 0)
        mask = to_attn_mask(attention_mask, False, B, S, St)
        enc, Sk = None, 0
        if encoder_hidden_states is not None:
            Sk = encoder_hidden_states.shape[1]
            enc = ops.convert(_f32_rows(encoder_hidden_states, "TransformerDecoder"), torch.bfloat16)  # once for all layers
        all_hidden_states = []
        current_key_values = []
        for i, layer_module in enumerate(self.layer):
            if return_hidden_states:
                all_hidden_states.append(x.view(B, S, d))
            # (the reference does not forward cross_attention_mask to its layers either: transformer.py:630-636)
            if caching:
                x, present = layer_module.run(x, B, S, mask, enc, Sk, past=past_key_values[i] if past_key_values is not None else None,
                                              use_cache=use_cache)
                if use_cache:
                    current_key_values.append(present)
            else:
                x = layer_module.run(x, B, S, mask, enc, Sk)
        x = x.view(B, S, d)
        if return_hidden_states:
            all_hidden_states.append(x)
        if self.final_layer_norm is not None:
            x = self.final_layer_norm(x)
        return TransformerOutput(last_hidden_state=x, hidden_states=all_hidden_states, current_key_values=current_key_values)


def _decoder_forward_train(self, hidden_states: Tensor, encoder_hidden_states, attention_mask, return_hidden_states: bool) -> TransformerOutput:
    """Differentiable TransformerDecoder.forward (DecoderStackFn: self-attention with the mask, optional cross-attention, feed-forward)."""
    return _decoder_layers_forward_train(list(self.layer), self.training, self.final_layer_norm, hidden_states, encoder_hidden_states,
                                         attention_mask, return_hidden_states)


def _decoder_layers_forward_train(dec_layers, training: bool, final_layer_norm, hidden_states: Tensor, encoder_hidden_states, attention_mask,
                                  return_hidden_states: bool, cross_attention_mask: Optional[Tensor] = None) -> TransformerOutput:
    """Differentiable forward of a list of TransformerDecoderLayers (a whole TransformerDecoder, or ONE stand-alone / wrapped layer)."""
    from ..._autograd import DecoderStackConfig, DecoderStackFn, draw_seed
    from ..._autograd import bias_or_zeros as _bias_or_zeros

    B, S, d = hidden_states.shape
    mask = to_attn_mask(attention_mask, False, B, S, S)
    layers, params = [], []
    bounds = [0]  # params[bounds[i]:bounds[i + 1]] belong to layer i
    drop_rates = set()
    for layer in dec_layers:
        # training-time dropout: the reference builds every dropout of a decoder layer from ONE value (:262-290) -- attention probabilities
        # (MultiHeadAttentionWithCache.dropout), the three residual branches, the MLP's hidden dropout
        rates = {float(layer.attention_dropout.p), float(layer.feedforward_dropout.p), float(layer.feedforward.hidden_dropout_p()),
                 float(layer.attention.dropout)}
        if layer.use_cross_attention and layer.cross_attention is not None:
            rates |= {float(layer.cross_attention_dropout.p), float(layer.cross_attention.dropout)}
        drop_rates.update(rates)
        steps = layer.feedforward.plan()
        if len(steps) != 2 or steps[1][1] != ops.ACT_NONE or steps[0][1] not in (ops.ACT_GELU_ERF, ops.ACT_QUICKGELU):
            raise ops.MmamdError("training: the feed-forward block must be Linear -> GELU/QuickGELU -> Linear")
        has_cross = bool(layer.use_cross_attention and encoder_hidden_states is not None)
        at = layer.attention
        bz = _bias_or_zeros  # (add_bias=False projections: a zero vector without grad stands in for the missing bias)
        params += [at.q_proj.weight, bz(at.q_proj), at.k_proj.weight, bz(at.k_proj), at.v_proj.weight, bz(at.v_proj),
                   at.output_proj.weight, at.output_proj.bias, layer.attention_layernorm.weight, layer.attention_layernorm.bias]
        spec = {"n_head": at.num_heads, "eps1": layer.attention_layernorm.eps, "eps2": layer.feedforward_layernorm.eps, "act": steps[0][1],
                "has_cross": has_cross, "post": not layer.norm_first}  # post-norm: the reference's default (transformer.py:289,435-470)
        if has_cross:
            ca = layer.cross_attention
            params += [ca.q_proj.weight, bz(ca.q_proj), ca.k_proj.weight, bz(ca.k_proj), ca.v_proj.weight, bz(ca.v_proj),
                       ca.output_proj.weight, ca.output_proj.bias, layer.cross_attention_layernorm.weight, layer.cross_attention_layernorm.bias]
            spec["epsc"] = layer.cross_attention_layernorm.eps
        params += [steps[0][0].weight, steps[0][0].bias, steps[1][0].weight, steps[1][0].bias, layer.feedforward_layernorm.weight,
                   layer.feedforward_layernorm.bias]
        layers.append(spec)
        bounds.append(len(params))
    enc2d, Sk = None, 0
    if encoder_hidden_states is not None:
        Sk = encoder_hidden_states.shape[1]
        e = encoder_hidden_states if encoder_hidden_states.is_contiguous() else encoder_hidden_states.contiguous()
        enc2d = e.view(B * Sk, e.shape[-1])
    cross_mask = to_attn_mask(cross_attention_mask, False, B, S, Sk) if (cross_attention_mask is not None and enc2d is not None) else None
    if training and len(drop_rates) > 1:
        raise ops.MmamdError(f"training: all dropout sites of a decoder stack must share one rate, got {sorted(drop_rates)}")
    drop_p = drop_rates.pop() if (drop_rates and training) else 0.0  # eval mode: every nn.Dropout is the identity
    seed = draw_seed() if drop_p > 0 else 0
    xc = hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous()
    all_hidden_states = []
    if return_hidden_states:
        # every hidden state attached to the graph like the reference's (:606-640): one autograd node per layer (layer0 keeps the dropout sites of the
        # one-node form, so both forms draw the same masks from the same seed)
        x = xc.view(B * S, d)
        for li in range(len(layers)):
            all_hidden_states.append(x.view(B, S, d))
            cfg = DecoderStackConfig(B, S, Sk, [layers[li]], mask, drop_p=drop_p, seed=seed, layer0=li, cross_mask=cross_mask)
            x = DecoderStackFn.apply(x, enc2d, cfg, *params[bounds[li]:bounds[li + 1]])
        x = x.view(B, S, d)
        all_hidden_states.append(x)
    else:
        cfg = DecoderStackConfig(B, S, Sk, layers, mask, drop_p=drop_p, seed=seed, cross_mask=cross_mask)
        x = DecoderStackFn.apply(xc.view(B * S, d), enc2d, cfg, *params).view(B, S, d)
    if final_layer_norm is not None:
        x = final_layer_norm(x)
    return TransformerOutput(last_hidden_state=x, hidden_states=all_hidden_states, current_key_values=[])


TransformerDecoder._forward_train = _decoder_forward_train
