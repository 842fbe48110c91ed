"""Host-side mirror of torchmultimodal/models/flava/transformer.py (TransformerEncoderLayer :77-221, TransformerEncoder
:224-293, FLAVATransformerWithoutEmbeddings :18-74, init_transformer_weights :296-310).

Same constructors, attribute names and construction order (state_dict keys `layer.N.attention.{query,key,value,output}`,
`layer.N.feedforward.model.{0,2}`, `layer.N.{attention,feedforward}_layernorm`; a seeded construction gives the
reference's initial weights).  One encoder layer on the MI355X is

    LN(fp32 -> bf16) -> in-proj GEMM [3d,d] -> attention (+mask, +probabilities) -> output GEMM (+bias, +residual, fp32)
    LN(fp32 -> bf16) -> up GEMM (+bias, GELU epilogue, bf16) -> down GEMM (+bias, +residual, fp32)

with the residual stream in fp32.  Every layer writes its output into a NEW buffer (residual read from the previous one),
so `hidden_states` — which FLAVA always asks for — are the live buffers, not copies.
"""
from __future__ import annotations

from functools import partial
from typing import Any, Callable, List, Optional, Tuple, Union

import torch
from torch import nn, Tensor

from ... import ops
from ..._packing import PackedCache
from ...modules.layers.attention import key_mask_from_attention_mask, MultiHeadAttention, SelfAttention
from ...modules.layers.mlp import MLP
from ...modules.layers.normalizations import Fp32LayerNorm
from ...modules.layers.transformer import TransformerOutput
from ..._autograd import plain_layers, wants_grad
from ..clip._transformer import forbid_training_forward


class FLAVATransformerWithoutEmbeddings(nn.Module):
    def __init__(
        self,
        encoder: nn.Module,
        layernorm: nn.Module,
        pooler: nn.Module,
        hidden_size: int = 768,
        weight_init_fn: Optional[Callable] = None,
        initializer_range: float = 0.02,
        use_cls_token: bool = True,
        **kwargs: Any,
    ):
        super().__init__()
        self.encoder = encoder
        self.layernorm = layernorm
        self.pooler = pooler
        if use_cls_token:
            self.cls_token = nn.Parameter(torch.zeros(1, 1, hidden_size))
        else:
            self.cls_token = None
        if weight_init_fn is None:
            weight_init_fn = partial(init_transformer_weights, initializer_range=initializer_range)
        self.apply(weight_init_fn)

    def forward(self, hidden_states: Optional[Tensor] = None, attention_mask: Optional[Tensor] = None) -> TransformerOutput:
        if hidden_states is None:
            raise ValueError("You have to specify hidden_states")
        if self.cls_token is not None:
            # [cls | tokens]: a strided device copy into one buffer (layout plumbing, no arithmetic)
            B, S, d = hidden_states.shape
            fused = torch.empty((B, S + 1, d), dtype=hidden_states.dtype, device=hidden_states.device)
            cls = self.cls_token if (torch.is_grad_enabled() and self.training) else self.cls_token.detach()
            fused[:, 0:1].copy_(cls.to(hidden_states.dtype).expand(B, -1, -1))  # differentiable w.r.t. cls_token in training
            fused[:, 1:].copy_(hidden_states)
            hidden_states = fused
        from ...schedule import get_schedule

        encoder_output = self.encoder(hidden_states, attention_mask=attention_mask, return_hidden_states=True,
                                      return_attn_weights=get_schedule().flava_attentions)
        sequence_output = self.layernorm(encoder_output.last_hidden_state)
        pooled_output = self.pooler(sequence_output) if self.pooler is not None else None
        return TransformerOutput(last_hidden_state=sequence_output, pooler_output=pooled_output,
                                 hidden_states=encoder_output.hidden_states, attentions=encoder_output.attentions)


class TransformerEncoderLayer(nn.Module):
    def __init__(
        self,
        d_model: int,
        n_head: int,
        dim_feedforward: int,
        dropout: float = 0.0,
        activation: Callable[..., nn.Module] = nn.ReLU,
        layer_norm_eps: float = 1e-12,
        norm_first: bool = False,
    ) -> None:
        super().__init__()
        self.attention = MultiHeadAttention(dim_q=d_model, dim_kv=d_model, n_head=n_head, attn_module=SelfAttention(dropout))
        self.attention_dropout = nn.Dropout(dropout)
        self.feedforward = MLP(d_model, d_model, dim_feedforward, dropout=dropout, activation=activation)
        self.feedforward_dropout = nn.Dropout(dropout)
        self.attention_layernorm = Fp32LayerNorm(d_model, eps=layer_norm_eps)
        self.feedforward_layernorm = Fp32LayerNorm(d_model, eps=layer_norm_eps)
        self.norm_first = norm_first
        self._packed = PackedCache()

    def _ln(self, ln: nn.LayerNorm, x: Tensor, out_dtype: torch.dtype) -> Tensor:
        pk = self._packed.get
        return ops.layernorm(x, pk(ln.weight, torch.float32), pk(ln.bias, torch.float32), ln.eps, out_dtype=out_dtype)

    def run(self, x: Tensor, B: int, S: int, key_mask: Optional[Tensor], want_probs: bool,
            head_mask: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
        """x: fp32 [B*S, d] (left untouched).  Returns (new fp32 [B*S, d], probabilities [B,H,S,S] or None)."""
        if self.training and (self.attention_dropout.p > 0 or self.feedforward_dropout.p > 0):
            raise ops.MmamdError("this non-differentiable (stand-alone / inference) forward applies no dropout: call .eval(); training-time dropout runs inside the encoder / decoder stacks' differentiable forwards")
        bf, f32 = torch.bfloat16, torch.float32
        if self.norm_first:  # reference :155-176
            hn = self._ln(self.attention_layernorm, x, bf)
            x1, probs = self.attention.run(hn, B, S, key_mask, want_probs, residual=x, head_mask=head_mask)
            hn = self._ln(self.feedforward_layernorm, x1, bf)
            y = self.feedforward.run(hn, residual=x1, out=x1)  # x1 is private to this layer: updated in place
            return y, probs
        # post-norm, reference :178-198
        a, probs = self.attention.run(ops.convert(x, bf), B, S, key_mask, want_probs, residual=x, head_mask=head_mask)
        x1 = self._ln(self.attention_layernorm, a, f32)
        ff = self.feedforward.run(ops.convert(x1, bf), residual=x1, out=a)
        return self._ln(self.feedforward_layernorm, ff, f32), probs

    def forward(self, hidden_states: Tensor, attention_mask: Optional[Tensor] = None, head_mask: Optional[Tensor] = None,
                return_attn_weights: bool = False) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        if wants_grad(self) or (torch.is_grad_enabled() and hidden_states.requires_grad):
            # differentiable stand-alone layer (a one-layer EncoderStackFn node, models/flava/_train.py): what a layer wrapped by FSDP /
            # checkpoint_wrapper (reference examples/flava/native/train.py:141-206) or a user's own loop over layers runs in training
            if hidden_states.dim() != 3 or hidden_states.dtype != torch.float32:
                raise ops.MmamdError("encoder layers on the MI355X path take fp32 [b, seq, c] hidden states in training")
            from ...schedule import get_schedule
            from ._train import run_layers

            B, S, _ = hidden_states.shape
            km = key_mask_from_attention_mask(attention_mask, B, S)
            y, _, probs = run_layers([self], self.training, hidden_states, km, False, return_attn_weights and get_schedule().train_attentions,
                                     head_mask=head_mask)
            return (y, probs[0] if probs is not None else None) if return_attn_weights else y
        shape = hidden_states.shape
        d = shape[-1]
        B = shape[0]
        S = hidden_states.numel() // (B * d)  # n-dimensional inputs [b, d1..dn, c] attend over the flattened positions
        xc = hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous()
        if xc.dtype != torch.float32:
            raise ops.MmamdError("encoder layers on the MI355X path take fp32 hidden states")
        km = key_mask_from_attention_mask(attention_mask, B, S)
        y, probs = self.run(xc.view(B * S, d), B, S, km, return_attn_weights, head_mask)
        y = y.view(shape)
        return (y, probs) if return_attn_weights else y


class TransformerEncoder(nn.Module):
    def __init__(
        self,
        n_layer: int,
        d_model: int,
        n_head: int,
        dim_feedforward: int,
        dropout: float = 0.0,
        activation: Callable[..., nn.Module] = nn.ReLU,
        layer_norm_eps: float = 1e-12,
        norm_first: bool = False,
        final_layer_norm_eps: Optional[float] = None,
    ):
        super().__init__()
        self.layer = nn.ModuleList([
            TransformerEncoderLayer(d_model, n_head, dim_feedforward, dropout, activation, layer_norm_eps, norm_first)
            for _ in range(n_layer)
        ])
        self.final_layer_norm = None
        if final_layer_norm_eps:
            self.final_layer_norm = Fp32LayerNorm(d_model, eps=final_layer_norm_eps)

    def forward(self, hidden_states: Tensor, attention_mask: Optional[Tensor] = None, head_mask: Optional[Tensor] = None,
                return_attn_weights: bool = False, return_hidden_states: bool = False) -> TransformerOutput:
        if hidden_states.dim() != 3 or hidden_states.dtype != torch.float32:
            raise ops.MmamdError("TransformerEncoder on the MI355X path takes fp32 [b, seq, c] hidden states")
        B, S, d = hidden_states.shape
        if not plain_layers(self.layer, TransformerEncoderLayer):
            # the reference's own loop (:254-290), every layer CALLED as a module: wrapped (FSDP, checkpoint_wrapper) or hooked layers
            all_hidden_states = [] if return_hidden_states else None
            all_self_attentions = [] if return_attn_weights else None
            x = hidden_states
            for layer_module in self.layer:
                if return_hidden_states:
                    all_hidden_states.append(x)
                out = layer_module(x, attention_mask=attention_mask, head_mask=head_mask, return_attn_weights=return_attn_weights)
                if return_attn_weights:
                    x, probs = out
                    all_self_attentions.append(probs)
                else:
                    x = out
            if return_hidden_states:
                all_hidden_states.append(x)
            if self.final_layer_norm is not None:
                x = self.final_layer_norm(x)
            if return_attn_weights and any(p is None for p in all_self_attentions):
                all_self_attentions = None  # (schedule.train_attentions = False)
            return TransformerOutput(last_hidden_state=x, hidden_states=all_hidden_states, attentions=all_self_attentions)
        if wants_grad(self) or (torch.is_grad_enabled() and hidden_states.requires_grad):
            # differentiable forward (models/flava/_train.py): every hidden state attached to the graph, attention probabilities as values
            # (schedule.train_attentions = False skips their recomputation: attentions = None, the r03 behaviour)
            from ...schedule import get_schedule
            from ._train import run_encoder

            km = key_mask_from_attention_mask(attention_mask, B, S)
            want_probs = return_attn_weights and get_schedule().train_attentions
            x, hidden, probs = run_encoder(self, hidden_states, km, return_hidden_states, want_probs, head_mask=head_mask)
            if self.final_layer_norm is not None:
                x = self.final_layer_norm(x)
            return TransformerOutput(last_hidden_state=x, hidden_states=hidden, attentions=probs)
        x = (hidden_states if hidden_states.is_contiguous() else hidden_states.contiguous()).view(B * S, d)
        km = key_mask_from_attention_mask(attention_mask, B, S)
        all_hidden_states: Optional[List[Tensor]] = [] if return_hidden_states else None
        all_self_attentions: Optional[List[Tensor]] = [] if return_attn_weights else None
        for layer_module in self.layer:
            if return_hidden_states:
                all_hidden_states.append(x.view(B, S, d))
            x, probs = layer_module.run(x, B, S, km, return_attn_weights, head_mask)  # (reference :268-275: the same head_mask for every layer)
            if return_attn_weights:
                all_self_attentions.append(probs)
        x = x.view(B, S, d)
        if return_hidden_states:
            all_hidden_states.append(x)
        if self.final_layer_norm is not None:
            x = self.final_layer_norm(x)
        return TransformerOutput(last_hidden_state=x, hidden_states=all_hidden_states, attentions=all_self_attentions)


def two_encoders_groupable(ea: "TransformerEncoder", eb: "TransformerEncoder") -> bool:
    """run_two_encoders applies: inference, pre-norm layers of equal depth, 64-wide heads, Linear -> GELU / QuickGELU -> Linear feed-forward
    blocks with the SAME activation (one grouped launch has one epilogue kind), no forward hooks on the layers."""
    if len(ea.layer) != len(eb.layer) or ea.final_layer_norm is not None or eb.final_layer_norm is not None:
        return False
    if not plain_layers(ea.layer, TransformerEncoderLayer) or not plain_layers(eb.layer, TransformerEncoderLayer):
        return False  # wrapped (FSDP / checkpoint) or hooked layers are called one by one
    acts = set()
    for enc in (ea, eb):
        for layer in enc.layer:
            at = layer.attention
            if (not layer.norm_first or layer._forward_hooks or layer._forward_pre_hooks or at.d_qk != 64 or at.key.in_features != at.query.in_features
                    or not isinstance(at.attn, SelfAttention) or at.query.bias is None):
                return False
            try:
                steps = layer.feedforward.plan()
            except ops.MmamdError:
                return False
            if len(steps) != 2 or steps[1][1] != ops.ACT_NONE or steps[0][1] not in (ops.ACT_GELU_ERF, ops.ACT_QUICKGELU):
                return False
            acts.add(steps[0][1])
    return len(acts) == 1


def run_two_encoders(ea: "TransformerEncoder", xa: Tensor, km_a: Optional[Tensor], eb: "TransformerEncoder", xb: Tensor, km_b: Optional[Tensor],
                     want_probs: bool = True):
    """Both towers of FLAVA's dual encoder LAYER-LOCKED on one stream (reference models/flava/model.py:127-205 runs them one after the other; they
    are independent until the multimodal encoder): per layer ONE grouped LayerNorm launch, ONE grouped persistent GEMM for each of the four
    projections over both towers' tiles (mmamd_gemm_bf16_grouped), and the two attention launches (probabilities, key-padding mask).  Same kernels'
    arithmetic per tower as TransformerEncoder.forward -> bit-identical outputs (tests/test_gpu_bench_size_parity.py).
    xa / xb: fp32 [B, S, d].  Returns ((x_L, hidden states, probabilities) per tower), every layer writing NEW residual-stream buffers (the hidden
    states FLAVA hands out are the live buffers, not copies)."""
    bf, f32 = torch.bfloat16, torch.float32
    (Ba, Sa, da), (Bb, Sb, db) = xa.shape, xb.shape
    xa2 = (xa if xa.is_contiguous() else xa.contiguous()).view(Ba * Sa, da)
    xb2 = (xb if xb.is_contiguous() else xb.contiguous()).view(Bb * Sb, db)
    dev = xa.device
    hid_a, hid_b, pr_a, pr_b = [xa2.view(Ba, Sa, da)], [xb2.view(Bb, Sb, db)], [], []

    def ln_pair(la, lb, ua, ub, oa, ob):
        pa, pb = la._packed.get, lb._packed.get
        na, nb = ua, ub
        ops.add_layernorm_grouped([(oa[0], None, pa(na.weight, f32), pa(na.bias, f32), na.eps, oa[1]),
                                   (ob[0], None, pb(nb.weight, f32), pb(nb.bias, f32), nb.eps, ob[1])])

    hna = torch.empty((Ba * Sa, da), dtype=bf, device=dev)
    hnb = torch.empty((Bb * Sb, db), dtype=bf, device=dev)
    qkva = torch.empty((Ba * Sa, 3 * da), dtype=bf, device=dev)
    qkvb = torch.empty((Bb * Sb, 3 * db), dtype=bf, device=dev)
    atta, attb = torch.empty_like(hna), torch.empty_like(hnb)
    ffa = ea.layer[0].feedforward.plan()[0][0].out_features
    ffb = eb.layer[0].feedforward.plan()[0][0].out_features
    upa = torch.empty((Ba * Sa, ffa), dtype=bf, device=dev)
    upb = torch.empty((Bb * Sb, ffb), dtype=bf, device=dev)
    for la, lb in zip(ea.layer, eb.layer):
        aa, ab = la.attention, lb.attention
        ca, cb = aa._packed, ab._packed
        ln_pair(la, lb, la.attention_layernorm, lb.attention_layernorm, (xa2, hna), (xb2, hnb))
        ops.gemm_bf16_grouped([(hna, ca.get_cat([aa.query.weight, aa.key.weight, aa.value.weight], bf),
                                ca.get_cat([aa.query.bias, aa.key.bias, aa.value.bias], f32), None, qkva),
                               (hnb, cb.get_cat([ab.query.weight, ab.key.weight, ab.value.weight], bf),
                                cb.get_cat([ab.query.bias, ab.key.bias, ab.value.bias], f32), None, qkvb)])
        for qkv, B, S, H, km, att, prs in ((qkva, Ba, Sa, aa.n_head, km_a, atta, pr_a), (qkvb, Bb, Sb, ab.n_head, km_b, attb, pr_b)):
            if want_probs or km is not None:
                _, probs = ops.attention_probs_fwd(qkv, B, S, H, km, want_probs=want_probs, out=att)
            else:
                ops.attention_fwd(qkv, B, S, H, causal=False, out=att)
                probs = None
            if want_probs:
                prs.append(probs)
        x1a, x1b = torch.empty_like(xa2), torch.empty_like(xb2)
        ops.gemm_bf16_grouped([(atta, ca.get(aa.output.weight, bf), ca.get(aa.output.bias, f32), xa2, x1a),
                               (attb, cb.get(ab.output.weight, bf), cb.get(ab.output.bias, f32), xb2, x1b)], out_dtype=f32)
        ln_pair(la, lb, la.feedforward_layernorm, lb.feedforward_layernorm, (x1a, hna), (x1b, hnb))
        (l1a, act), (l2a, _) = la.feedforward.plan()
        (l1b, _), (l2b, _) = lb.feedforward.plan()
        fa, fb = la.feedforward._packed.get, lb.feedforward._packed.get
        ops.gemm_bf16_grouped([(hna, fa(l1a.weight, bf), fa(l1a.bias, f32), None, upa), (hnb, fb(l1b.weight, bf), fb(l1b.bias, f32), None, upb)],
                              act=act)
        ops.gemm_bf16_grouped([(upa, fa(l2a.weight, bf), fa(l2a.bias, f32), x1a, x1a), (upb, fb(l2b.weight, bf), fb(l2b.bias, f32), x1b, x1b)],
                              out_dtype=f32)
        xa2, xb2 = x1a, x1b
        hid_a.append(xa2.view(Ba, Sa, da))
        hid_b.append(xb2.view(Bb, Sb, db))
    return (hid_a[-1], hid_a, pr_a if want_probs else None), (hid_b[-1], hid_b, pr_b if want_probs else None)


def init_transformer_weights(module: nn.Module, initializer_range: float) -> None:
    """Initialize the weights (reference :296-310)."""
    if isinstance(module, (nn.Linear, nn.Conv2d)):
        module.weight.data.normal_(mean=0.0, std=initializer_range)
        if module.bias is not None:
            module.bias.data.zero_()
    elif isinstance(module, nn.Embedding):
        module.weight.data.normal_(mean=0.0, std=initializer_range)
        if module.padding_idx is not None:
            module.weight.data[module.padding_idx].zero_()
    elif isinstance(module, nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)
