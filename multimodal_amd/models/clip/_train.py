"""Training step of the CLIP towers on the MI355X kernels: autograd nodes whose forward AND
backward are HIP kernels behind the C-ABI.  What torch autograd does for the reference (nn.TransformerEncoderLayer,
nn.MultiheadAttention, LayerNorm, Conv2d patch embedding, embeddings, the pooled projections, F.normalize) is restated as:

    forward (per layer, tensors kept for backward in brackets):
        [x] -LN-> [h1] -GEMM-> [qkv] -attention-> [att, lse] -GEMM(+x)-> [x_mid] -LN-> [h2] -GEMM-> [u] -act-> [g] -GEMM(+x_mid)-> x'
    backward:  dgrad  dX = dY W        = gemm_bf16(dY, W^T)                    (W^T: cached bf16 transpose)
               wgrad  dW = dY^T X      = gemm_bf16(dY^T, X^T)  contraction = token index (operands transposed by a kernel)
               bias   db = column sums;  LayerNorm / activation / attention / normalize: their own backward kernels

The residual-stream gradient stays fp32; everything that feeds an MFMA is bf16 (same rule as the forward).
"""
from __future__ import annotations

from typing import List, Tuple

import torch
from torch import Tensor

from ... import ops

bf, f32 = torch.bfloat16, torch.float32

_LAYER_PARAMS = ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
                 "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight",
                 "norm2.bias")


def _get(mod, dotted):
    for part in dotted.split("."):
        mod = getattr(mod, part)
    return mod


def _c32(t: Tensor) -> Tensor:
    t = t.detach()
    if t.dtype != f32:
        raise ops.MmamdError("training on the MI355X path keeps parameters in float32")
    return t if t.is_contiguous() else t.contiguous()


def _dgrad(dy: Tensor, w: Tensor, out_dtype, act: int = ops.ACT_NONE, pre_act: Tensor = None) -> Tensor:
    """dX[M,K] = dY[M,N] . W[N,K]  (W fp32 [N,K], N % 64 == 0); with act = ACT_MUL_*_GRAD the epilogue multiplies by
    act'(pre_act): the activation's backward without a pass of its own."""
    N = w.shape[0]
    if N % 64 != 0:
        raise ops.MmamdError(f"backward GEMM: output width {N} of a Linear must be a multiple of 64")
    wT = ops.transpose_to_bf16(w, pad_to=64)  # bf16 [K, N]
    return ops.gemm_bf16(dy, wT, None, act=act, residual=pre_act, out_dtype=out_dtype)


def _wgrad(dy: Tensor, x: Tensor, bias: bool = False):
    """dW[N,K] = dY^T X for dy [M,N], x [M,K] (bf16 or fp32 row-major); contraction over the M tokens (zero-padded to 128).
    bias=True also returns db[N] = column sums of dY, produced by the same transpose pass over dY."""
    if bias:
        dyT, db = ops.transpose_to_bf16(dy, with_colsum=True)  # [N, Mp]
    else:
        dyT, db = ops.transpose_to_bf16(dy), None
    xT = ops.transpose_to_bf16(x)    # [K, Mp]
    dW = ops.gemm_bf16_splitk(dyT, xT)
    return (dW, db) if bias else dW


class StackFn(torch.autograd.Function):
    """x0 fp32 [B*S, d] -> x_L through the pre-norm layers of a TransformerStack (models/clip/_transformer.py)."""

    @staticmethod
    def forward(ctx, x0: Tensor, stack, B: int, S: int, causal: bool, *params: Tensor):
        H, d = stack.nhead, stack.d_model
        saved: List[Tensor] = []
        x = x0.detach()
        for li, layer in enumerate(stack.layers):
            Wqkv, bqkv, Wo, bo, W1, b1, W2, b2, g1, be1, g2, be2 = (_c32(p) for p in params[12 * li:12 * li + 12])
            h1 = ops.layernorm(x, g1, be1, layer.norm1.eps, out_dtype=bf)
            qkv = ops.gemm_bf16(h1, ops.convert(Wqkv, bf), bqkv)
            att, lse = ops.attention_fwd_train(qkv, B, S, H, causal)
            x_mid = ops.gemm_bf16(att, ops.convert(Wo, bf), bo, residual=x, out_dtype=f32,
                                  out=torch.empty_like(x))
            h2 = ops.layernorm(x_mid, g2, be2, layer.norm2.eps, out_dtype=bf)
            u = ops.gemm_bf16(h2, ops.convert(W1, bf), b1)
            g = ops.act_fwd(u, ops.ACT_QUICKGELU)
            x_out = ops.gemm_bf16(g, ops.convert(W2, bf), b2, residual=x_mid, out_dtype=f32, out=torch.empty_like(x))
            saved += [x, h1, qkv, att, lse, x_mid, h2, u, g]
            x = x_out
        ctx.save_for_backward(*saved, *[p for p in params])
        ctx.meta = (stack, B, S, causal, len(params))
        return x

    @staticmethod
    def backward(ctx, dx_out: Tensor):
        stack, B, S, causal, nparam = ctx.meta
        tensors = ctx.saved_tensors
        saved, params = tensors[:len(tensors) - nparam], tensors[len(tensors) - nparam:]
        H = stack.nhead
        dX = dx_out.detach()
        dX = dX if dX.is_contiguous() else dX.contiguous()
        grads: List[Tensor] = [None] * nparam
        dXb = None  # bf16 copy of dX: produced by the LayerNorm backward of the layer above
        for li in reversed(range(len(stack.layers))):
            layer = stack.layers[li]
            x, h1, qkv, att, lse, x_mid, h2, u, g = saved[9 * li:9 * li + 9]
            Wqkv, bqkv, Wo, bo, W1, b1, W2, b2, g1, be1, g2, be2 = (_c32(p) for p in params[12 * li:12 * li + 12])
            if dXb is None:
                dXb = ops.convert(dX, bf)
            # x_out = x_mid + g W2^T + b2;  g = QuickGELU(u): du = (dX W2) * act'(u) in the dgrad GEMM's epilogue
            du = _dgrad(dXb, W2, bf, ops.ACT_MUL_QUICKGELU_GRAD, u)
            dW2, db2 = _wgrad(dXb, g, bias=True)
            # u = h2 W1^T + b1
            dh2 = _dgrad(du, W1, f32)
            dW1, db1 = _wgrad(du, h2, bias=True)
            dx_mid, dg2, dbe2, dxmb = ops.layernorm_bwd(x_mid, g2, dh2, layer.norm2.eps, add=dX, want_bf16=True)
            # x_mid = x + att Wo^T + bo
            datt = _dgrad(dxmb, Wo, bf)
            dWo, dbo = _wgrad(dxmb, att, bias=True)
            dqkv = ops.attention_bwd(qkv, att, datt, lse, B, S, H, causal)
            # qkv = h1 Wqkv^T + bqkv
            dh1 = _dgrad(dqkv, Wqkv, f32)
            dWqkv, dbqkv = _wgrad(dqkv, h1, bias=True)
            dX, dg1, dbe1, dXb = ops.layernorm_bwd(x, g1, dh1, layer.norm1.eps, add=dx_mid, want_bf16=True)
            grads[12 * li:12 * li + 12] = [dWqkv, dbqkv, dWo, dbo, dW1, db1, dW2, db2, dg1, dbe1, dg2, dbe2]
        return (dX, None, None, None, None, *grads)


def run_stack(stack, x0: Tensor, B: int, S: int, causal: bool) -> Tensor:
    params = [_get(layer, n) for layer in stack.layers for n in _LAYER_PARAMS]
    return StackFn.apply(x0, stack, B, S, causal, *params)


class VisionEmbedFn(torch.autograd.Function):
    """images -> fp32 residual stream [B*(G2+1), w]: conv patch embedding, CLS, + positional embedding, ln_pre."""

    @staticmethod
    def forward(ctx, images, conv_w, cls, pos, ln_w, ln_b, patch: int, eps: float):
        B = images.shape[0]
        w = conv_w.shape[0]
        K = conv_w.shape[1] * patch * patch
        kpad = (K + 63) // 64 * 64
        cols = ops.patchify(images if images.is_contiguous() else images.contiguous(), patch, kpad)  # bf16 [B*G2, kpad]
        wk = torch.zeros((w, kpad), dtype=bf, device=images.device)
        wk[:, :K].copy_(ops.convert(_c32(conv_w).view(w, K), bf))
        pe = ops.gemm_bf16(cols, wk, None, out_dtype=f32)
        G2 = cols.shape[0] // B
        asm = ops.flava_image_embed(pe, _c32(cls).view(-1), _c32(pos), B, G2)  # cls + pos[0] | pe + pos[1:]
        x0 = ops.layernorm(asm, _c32(ln_w), _c32(ln_b), eps, out_dtype=f32)
        ctx.save_for_backward(cols, asm, ln_w)
        ctx.meta = (B, G2, w, K, kpad, tuple(conv_w.shape), eps, tuple(cls.shape), tuple(pos.shape))
        return x0

    @staticmethod
    def backward(ctx, dx0):
        cols, asm, ln_w = ctx.saved_tensors
        B, G2, w, K, kpad, conv_shape, eps, cls_shape, pos_shape = ctx.meta
        S = G2 + 1
        d_asm, dg, db = ops.layernorm_bwd(asm, _c32(ln_w), dx0.contiguous(), eps)
        dpos = ops.colsum(d_asm.view(B, S * w)).view(S, w)  # asm[b, s] = (...) + pos[s]
        dcls = dpos[0].clone()                               # asm[b, 0] = cls + pos[0]: the same sum over the batch
        idx = (torch.arange(B * S, device=dx0.device, dtype=torch.int32).view(B, S)[:, 1:]).reshape(-1).contiguous()
        d_pe = ops.gather_rows(d_asm, w, idx, w, bf)        # rows of the patch tokens, bf16 [B*G2, w]
        dwk = _wgrad(d_pe, cols)                            # [w, kpad]
        dconv = dwk[:, :K].contiguous().view(conv_shape)
        return None, dconv, dcls.view(cls_shape), dpos.view(pos_shape), dg, db, None, None


class TextEmbedFn(torch.autograd.Function):
    """ids -> fp32 residual stream: token_embedding[ids] + positional_embedding."""

    @staticmethod
    def forward(ctx, ids, table, pos):
        x0 = ops.embed_tokens(ids, _c32(table), _c32(pos))
        ctx.save_for_backward(ids)
        ctx.meta = (tuple(table.shape), tuple(pos.shape))
        return x0

    @staticmethod
    def backward(ctx, dx0):
        (ids,) = ctx.saved_tensors
        tshape, pshape = ctx.meta
        B, S = ids.shape
        d = tshape[1]
        dx0 = dx0.contiguous()
        dpos = ops.colsum(dx0.view(B, S * d)).view(S, d)
        dtable = torch.zeros(tshape, dtype=f32, device=dx0.device)  # memset; rows collide -> fp32 atomics
        ops.scatter_add_rows_(dtable, ids.reshape(-1).contiguous(), dx0)
        dpos_full = dpos if pshape[0] == S else torch.cat([dpos, torch.zeros((pshape[0] - S, d), dtype=f32, device=dx0.device)])
        return None, dtable, dpos_full.view(pshape)


class PooledHeadFn(torch.autograd.Function):
    """x_L fp32 [B*S, d], row index per sample -> LayerNorm(row) @ P   (ln_post + projection / ln_final + EOT + projection)."""

    @staticmethod
    def forward(ctx, x, idx64, ln_w, ln_b, proj, eps: float, proj_is_linear: bool):
        B = idx64.numel()
        d = x.shape[1]
        rows = ops.gather_rows(x, d, idx64.to(torch.int32), d, f32)
        n = ops.layernorm(rows, _c32(ln_w), _c32(ln_b), eps, out_dtype=f32)
        P = _c32(proj)
        E = P.shape[0] if proj_is_linear else P.shape[1]
        e = ops.f32_gemm_strided(n, d, 1, P, d if proj_is_linear else 1, 1 if proj_is_linear else E, B, E, d)
        ctx.save_for_backward(rows, n, idx64, ln_w, proj)
        ctx.meta = (eps, proj_is_linear, tuple(x.shape), E)
        return e

    @staticmethod
    def backward(ctx, de):
        rows, n, idx64, ln_w, proj = ctx.saved_tensors
        eps, lin, xshape, E = ctx.meta
        B, d = rows.shape
        de = de.contiguous()
        P = _c32(proj)
        if lin:   # P [E, d]: dP[j, k] = sum_b de[b, j] n[b, k];  dn[b, k] = sum_j de[b, j] P[j, k]
            dP = ops.f32_gemm_strided(de, 1, E, n, 1, d, E, d, B)
            dn = ops.f32_gemm_strided(de, E, 1, P, 1, d, B, d, E)
        else:     # P [d, E]: dP[k, j] = sum_b n[b, k] de[b, j];  dn[b, k] = sum_j de[b, j] P[k, j]
            dP = ops.f32_gemm_strided(n, 1, d, de, 1, E, d, E, B)
            dn = ops.f32_gemm_strided(de, E, 1, P, E, 1, B, d, E)
        drows, dg, db = ops.layernorm_bwd(rows, _c32(ln_w), dn, eps)
        dx = torch.zeros(xshape, dtype=f32, device=de.device)  # memset: only the pooled rows receive gradient
        ops.scatter_add_rows_(dx, idx64, drows)
        return dx, None, dg, db, dP, None, None


class L2NormalizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xc = _c32(x)
        ctx.save_for_backward(xc)
        return ops.l2_normalize(xc)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.l2_normalize_bwd(x, dy.contiguous())


def wants_grad(module) -> bool:
    return module.training and torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())
