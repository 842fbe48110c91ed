"""Training (differentiable) forward of the FLAVA encoders on the MI355X kernels: the autograd nodes that are specific to FLAVA;
the layer stack, LayerNorm, CLS projections and normalisation nodes are the shared ones of multimodal_amd/_autograd.py.

Differentiable: ImageEmbeddings, BERTTextEmbeddings, flava.TransformerEncoder (pre-norm, GELU / QuickGELU, key-padding masks),
Fp32LayerNorm, the CLS projections, the token-wise mm projections, FLAVAGlobalContrastiveLoss.  Not differentiable (forward-only
outputs, returned detached): attention probabilities (not produced in training mode), Pooler outputs, the MLM / MIM / ITM heads.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn, Tensor

from ... import ops
from ..._autograd import EncoderStackFn, StackConfig, bias_or_zeros, c32, dgrad, stack_drop_spec, wgrad

bf, f32 = torch.bfloat16, torch.float32


class FlavaImageEmbedFn(torch.autograd.Function):
    """pixel_values -> fp32 [B, G2+1, d]: conv patch embedding (+bias), optional mask-token blend (binary patch mask), CLS,
    + position embeddings (models/flava/image_encoder.py:139-177)."""

    @staticmethod
    def forward(ctx, images, conv_w, conv_b, cls, pos, patch: int, patches_mask, mask_token):
        B = images.shape[0]
        w = conv_w.shape[0]
        K = conv_w.shape[1] * patch * patch
        kpad = (K + 63) // 64 * 64  # e.g. patch 14: 588 -> 640 (zero columns on both operands)
        cols = ops.patchify(images if images.is_contiguous() else images.contiguous(), patch, kpad)
        wk = ops.convert(c32(conv_w).view(w, K), bf)
        if kpad != K:
            wp = torch.zeros((w, kpad), dtype=bf, device=images.device)
            wp[:, :K].copy_(wk)
            wk = wp
        pe = ops.gemm_bf16(cols, wk, c32(conv_b), out_dtype=f32)
        G2 = cols.shape[0] // B
        pm = None
        if patches_mask is not None and mask_token is not None:
            pm = patches_mask.reshape(B, G2)
            pm = (pm if pm.dtype == torch.int64 else pm.to(torch.int64)).contiguous()
        hc = 1 if cls is not None else 0  # no CLS row: CoCa's ViT (layers/patch_embedding.py, include_cls_embed=False)
        x = ops.flava_image_embed(pe, c32(cls).view(-1) if hc else None, c32(pos).view(G2 + hc, w), B, G2, pm,
                                  c32(mask_token).view(-1) if pm is not None else None)
        ctx.save_for_backward(cols, pm if pm is not None else torch.empty(0, device=images.device))
        ctx.meta = (B, G2, w, tuple(conv_w.shape), tuple(cls.shape) if hc else None, tuple(pos.shape), pm is not None,
                    tuple(mask_token.shape) if mask_token is not None else None, K)
        return x.view(B, G2 + hc, w)

    @staticmethod
    def backward(ctx, dx):
        cols, pm = ctx.saved_tensors
        B, G2, w, conv_shape, cls_shape, pos_shape, masked, mt_shape, K = ctx.meta
        hc = 1 if cls_shape is not None else 0
        S = G2 + hc
        d_asm = dx.detach().contiguous().view(B * S, w)
        dpos = ops.colsum(d_asm.view(B, S * w)).view(S, w)
        dcls = dpos[0].clone() if hc else None
        idx = (torch.arange(B * S, device=dx.device, dtype=torch.int32).view(B, S)[:, hc:]).reshape(-1).contiguous()
        # masked patches (w = 1) take the mask token instead of their embedding: zero gradient for the embedding there
        d_pe = ops.gather_rows(d_asm, w, idx, w, bf, zero_rows=pm.view(-1) if masked else None)
        dW, db = wgrad(d_pe, cols, bias=True)
        dmt = None
        if masked:  # sum over the masked patch rows = (sum over all patch rows) - (sum over the unmasked ones = the conv-bias gradient)
            dmt = (dpos[hc:].sum(0) - db).view(mt_shape)
        elif mt_shape is not None:
            dmt = torch.zeros(mt_shape, dtype=f32, device=dx.device)
        dconv = (dW if dW.shape[1] == K else dW[:, :K].contiguous()).view(conv_shape)
        return None, dconv, db, dcls.view(cls_shape) if hc else None, dpos.view(pos_shape), None, None, dmt


class BicubicTableFn(torch.autograd.Function):
    """ImageEmbeddings.interpolate_pos_encoding (models/flava/image_encoder.py:102-137) as a differentiable map of the position table: the
    bicubic resampling of the patch grid (CLS row passed through) is LINEAR in the table, so its backward is the transpose of the same map.  The
    map's matrix A [1 + h0 w0, 1 + n] is what the forward kernel makes of an identity table; d table = A^T d out in exact fp32
    (mmamd_f32_gemm_strided).  r05: interpolate_pos_encoding raised in training."""

    @staticmethod
    def forward(ctx, table, h0: int, w0: int, scale_h: float, scale_w: float):
        t = c32(table)
        out = ops.bicubic_pos_embed(t.view(t.shape[-2], t.shape[-1]), h0, w0, scale_h, scale_w)
        ctx.meta = (tuple(table.shape), h0, w0, scale_h, scale_w)
        return out.view(1, out.shape[0], out.shape[1])

    @staticmethod
    def backward(ctx, dout):
        shape, h0, w0, scale_h, scale_w = ctx.meta
        n1, d = shape[-2], shape[-1]
        eye = torch.eye(n1, dtype=f32, device=dout.device)
        A = ops.bicubic_pos_embed(eye, h0, w0, scale_h, scale_w)  # [1 + h0 w0, n1]
        g = dout.detach().contiguous().view(A.shape[0], d)
        # dtable[j, k] = sum_i A[i, j] g[i, k]:  X(m = j, k = i) = A[i n1 + j],  Y(n = k, k = i) = g[i d + k]
        dt = ops.f32_gemm_strided(A, 1, n1, g, 1, d, n1, d, A.shape[0])
        return dt.view(shape), None, None, None, None


class BertEmbedFn(torch.autograd.Function):
    """LayerNorm(word[ids] + position[pos] + token_type[type]) (modules/layers/text_embedding.py:74-104)."""

    @staticmethod
    def forward(ctx, ids, word, pos, typ, ln_w, ln_b, eps: float, token_type_ids, position_ids, padding_idx):
        B, S = ids.shape
        e = ops.bert_embed_ln(ids, c32(word), c32(pos), c32(typ), None, None, eps, token_type_ids, position_ids)  # un-normalised sum
        x = ops.layernorm(e, c32(ln_w), c32(ln_b), eps, out_dtype=f32)
        ctx.save_for_backward(e, ids, ln_w, token_type_ids if token_type_ids is not None else torch.empty(0, device=ids.device),
                              position_ids if position_ids is not None else torch.empty(0, device=ids.device))
        ctx.meta = (eps, tuple(word.shape), tuple(pos.shape), tuple(typ.shape), token_type_ids is not None, position_ids is not None, padding_idx)
        return x.view(B, S, -1)

    @staticmethod
    def backward(ctx, dx):
        e, ids, ln_w, tt, pid = ctx.saved_tensors
        eps, wshape, pshape, tshape, has_tt, has_pid, padding_idx = ctx.meta
        B, S = ids.shape
        d = wshape[1]
        de, dg, db = ops.layernorm_bwd(e, c32(ln_w), dx.detach().contiguous().view(B * S, d), eps)
        dev = de.device
        dword = torch.zeros(wshape, dtype=f32, device=dev)  # memset; rows collide -> fp32 atomics
        ops.scatter_add_rows_(dword, ids.reshape(-1).contiguous(), de)
        if padding_idx is not None:
            dword[padding_idx].zero_()  # nn.Embedding(padding_idx=...): the pad row receives no gradient
        dpos = torch.zeros(pshape, dtype=f32, device=dev)
        if has_pid:
            ops.scatter_add_rows_(dpos, pid.reshape(-1).contiguous(), de)
        else:
            dpos[:S].copy_(ops.colsum(de.view(B, S * d)).view(S, d))  # position s is shared by the B samples
        dtyp = torch.zeros(tshape, dtype=f32, device=dev)
        if has_tt:
            ops.scatter_add_rows_(dtyp, tt.reshape(-1).contiguous(), de)
        else:
            dtyp[0].copy_(ops.colsum(de))                              # every token has type 0
        return None, dword, dpos, dtyp, dg, db, None, None, None, None


class TokenLinearFn(torch.autograd.Function):
    """y = x W^T + b over every token of a [B, S, d] fp32 tensor (image_to_mm / text_to_mm projections, models/flava/model.py:294-295)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, S, d = x.shape
        xb = ops.convert((x.detach() if x.is_contiguous() else x.detach().contiguous()).view(B * S, d), bf)
        y = ops.gemm_bf16(xb, ops.convert(c32(weight), bf), c32(bias) if bias is not None else None, out_dtype=f32)
        ctx.save_for_backward(xb, weight)
        ctx.meta = (B, S, bias is not None)
        return y.view(B, S, -1)

    @staticmethod
    def backward(ctx, dy):
        xb, weight = ctx.saved_tensors
        B, S, has_bias = ctx.meta
        dyb = ops.convert(dy.detach().contiguous().view(B * S, -1), bf)
        dx = dgrad(dyb, c32(weight), f32)
        if has_bias:
            dW, db = wgrad(dyb, xb, bias=True)
        else:
            dW, db = wgrad(dyb, xb), None
        return dx.view(B, S, -1), dW, db


_FLAVA_LAYER_PARAMS = ("attention.query.weight", "attention.query.bias", "attention.key.weight", "attention.key.bias",
                       "attention.value.weight", "attention.value.bias", "attention.output.weight", "attention.output.bias",
                       "ff0.weight", "ff0.bias", "ff1.weight", "ff1.bias", "attention_layernorm.weight", "attention_layernorm.bias",
                       "feedforward_layernorm.weight", "feedforward_layernorm.bias")


def _to_canonical(p: List[Tensor]):
    qw, qb, kw, kb, vw, vb, ow, ob, w1, b1, w2, b2, g1, be1, g2, be2 = p
    return [torch.cat([qw, kw, vw], 0), torch.cat([qb, kb, vb], 0), ow, ob, w1, b1, w2, b2, g1, be1, g2, be2]  # stacking: copies


def _from_canonical(g: List[Tensor]):
    dWqkv, dbqkv, dWo, dbo, dW1, db1, dW2, db2, dg1, dbe1, dg2, dbe2 = g
    d = dWqkv.shape[1]
    return [dWqkv[:d], dbqkv[:d], dWqkv[d:2 * d], dbqkv[d:2 * d], dWqkv[2 * d:], dbqkv[2 * d:], dWo, dbo, dW1, db1, dW2, db2, dg1, dbe1,
            dg2, dbe2]


def run_encoder(encoder, x: Tensor, key_mask: Optional[Tensor], keep_hidden: bool, want_probs: bool = False, head_mask: Optional[Tensor] = None):
    """Differentiable pass through a flava.TransformerEncoder (all its layers as ONE autograd node)."""
    return run_layers(list(encoder.layer), encoder.training, x, key_mask, keep_hidden, want_probs, head_mask=head_mask)


def run_layers(layers, training: bool, x: Tensor, key_mask: Optional[Tensor], keep_hidden: bool, want_probs: bool = False,
               head_mask: Optional[Tensor] = None):
    """Differentiable pass through a list of flava TransformerEncoderLayers (a whole encoder, or ONE stand-alone / wrapped layer).  Returns (x_L [B,S,d], hidden states or None, attention probabilities or None).
    keep_hidden: ALL hidden states, attached to the graph (the input, the input of every further layer, the result) like the reference's
    training forward (models/flava/transformer.py:254-259).  want_probs: the per-layer attention probabilities [B,H,S,S] fp32, recomputed from
    each layer's saved projections (and, unmasked, its saved log-sum-exp: mmamd_attention_probs_from_lse; else the inference kernel
    mmamd_attention_probs_fwd) -- values as in eval mode, NOT differentiable (the
    reference's are; nothing in its models or losses differentiates through returned attention maps).
    head_mask (reference layers/attention.py:236-237; the same mask for every layer, flava/transformer.py:268-275): multiplied into the probabilities
    after softmax in the forward and the backward kernels (a constant: no gradient of its own); the returned maps carry it too."""
    from ...modules.layers.mlp import fused_activation_code

    B, S, d = x.shape
    params, eps1, eps2, act = [], [], [], None
    norm_first = {bool(layer.norm_first) for layer in layers}
    if len(norm_first) != 1:
        raise ops.MmamdError("training: all layers of a stack must share norm_first")
    for layer in layers:
        steps = layer.feedforward.plan()
        if len(steps) != 2 or steps[1][1] != ops.ACT_NONE or steps[0][1] not in (ops.ACT_GELU_ERF, ops.ACT_QUICKGELU):
            raise ops.MmamdError("training: the feed-forward block must be Linear -> GELU/QuickGELU -> Linear")
        act = steps[0][1] if act is None else act
        if act != steps[0][1]:
            raise ops.MmamdError("training: all layers of a stack must use the same activation")
        at = layer.attention
        params += [at.query.weight, bias_or_zeros(at.query), at.key.weight, bias_or_zeros(at.key), at.value.weight, bias_or_zeros(at.value), at.output.weight,
                   at.output.bias, steps[0][0].weight, steps[0][0].bias, steps[1][0].weight, steps[1][0].bias,
                   layer.attention_layernorm.weight, layer.attention_layernorm.bias, layer.feedforward_layernorm.weight,
                   layer.feedforward_layernorm.bias]
        eps1.append(layer.attention_layernorm.eps)
        eps2.append(layer.feedforward_layernorm.eps)
    # training-time dropout (reference flava/transformer.py: attention_dropout / feedforward_dropout on the branches, the MLP's hidden dropout,
    # and SelfAttention(attn_dropout) on the attention probabilities -- the general attention kernels then carry the Philox mask)
    drop, seed = stack_drop_spec(layers, attn_p=lambda l: l.attention.attn.attn_dropout, training=training)
    hm = None
    if head_mask is not None:
        from ...modules.layers.attention import head_mask_f32

        hm = head_mask_f32(head_mask)
    cfg = StackConfig(len(layers), layers[0].attention.n_head, B, S, False, act, eps1, eps2, 16, _to_canonical,
                      _from_canonical, key_mask=key_mask, keep_hidden=keep_hidden, drop=drop, seed=seed, norm_first=norm_first.pop(), head_mask=hm)
    cfg.keep_hidden = keep_hidden or want_probs
    xc = x if x.is_contiguous() else x.contiguous()
    res = EncoderStackFn.apply(xc.view(B * S, d), cfg, *params)
    y = (res[0] if cfg.keep_hidden else res).view(B, S, d)
    hidden = [x] + [h.view(B, S, d) for h in res[1:]] + [y] if keep_hidden else None
    probs = None
    if want_probs:
        H = cfg.n_head
        if hm is not None:  # the masked maps, as the inference path returns them: one general-attention pass per layer with the probabilities written
            d3 = cfg.qkv[0].shape[1] // 3
            probs = [ops.attention_x_fwd(q[:, :d3], q[:, d3:2 * d3], q[:, 2 * d3:], B, S, S, H, d3 // H, ops.AttnMask(key_mask=key_mask),
                                         want_probs=True, head_mask=hm)[1] for q in cfg.qkv]
        elif key_mask is None and ops.attention_probs_from_lse_supported(S) and len(cfg.lse) == len(cfg.qkv):
            # one pass per layer from the saved projections and log-sum-exp rows: exp2(scale q.k - lse), no second attention
            probs = [ops.attention_probs_from_lse(q, l, B, S, H) for q, l in zip(cfg.qkv, cfg.lse)]
        else:
            probs = [ops.attention_probs_fwd(q, B, S, H, key_mask)[1] for q in cfg.qkv]
    return y, hidden, probs
