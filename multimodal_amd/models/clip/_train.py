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
from ..._autograd import EncoderStackFn, L2NormalizeFn, StackConfig, c32 as _c32, grad_requested as wants_grad, wgrad as _wgrad  # noqa: F401
from ..._custom_op import define as _define
# (CLIP: the differentiable path returns exactly what the inference path returns, so it is taken whenever autograd would record
#  the call in the reference -- train OR eval mode; `wants_grad(module, *inputs)`)

bf, f32 = torch.bfloat16, torch.float32

_LAYER_PARAMS = ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
                 "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight",
                 "norm2.bias")


def _get(mod, dotted):
    for part in dotted.split("."):
        mod = getattr(mod, part)
    return mod


def run_stack(stack, x0: Tensor, B: int, S: int, causal: bool) -> Tensor:
    """The pre-norm layers of a TransformerStack (models/clip/_transformer.py): torch's parameter layout IS the canonical one."""
    cfg, params = _stack_cfg(stack, B, S, causal)
    return EncoderStackFn.apply(x0, cfg, *params)


def _stack_cfg(stack, B: int, S: int, causal: bool):
    params = [_get(layer, n) for layer in stack.layers for n in _LAYER_PARAMS]
    ident = lambda t: t
    return StackConfig(len(stack.layers), stack.nhead, B, S, causal, ops.ACT_QUICKGELU, [l.norm1.eps for l in stack.layers],
                       [l.norm2.eps for l in stack.layers], 12, ident, ident), params


def _vision_embed_fwd(images: Tensor, conv_w: Tensor, cls: Tensor, pos: Tensor, ln_w: Tensor, ln_b: Tensor, patch: int, eps: float) -> List[Tensor]:
    """-> [x0 fp32 [B*(G2+1), w], cols (bf16 im2col rows), asm (the residual stream before ln_pre)]"""
    B = images.shape[0]
    w = conv_w.shape[0]
    K = conv_w.shape[1] * patch * patch
    kpad = (K + 63) // 64 * 64
    cols = ops.patchify(images if images.is_contiguous() else images.contiguous(), patch, kpad)  # bf16 [B*G2, kpad]
    wk = torch.zeros((w, kpad), dtype=bf, device=images.device)
    wk[:, :K].copy_(ops.convert(conv_w.view(w, K), bf))
    pe = ops.gemm_bf16(cols, wk, None, out_dtype=f32)
    G2 = cols.shape[0] // B
    asm = ops.flava_image_embed(pe, cls.view(-1), pos, B, G2)  # cls + pos[0] | pe + pos[1:]
    x0 = ops.layernorm(asm, ln_w, ln_b, eps, out_dtype=f32)
    return [x0, cols, asm]


def _vision_embed_fwd_fake(images, conv_w, cls, pos, ln_w, ln_b, patch, eps):
    B, w = images.shape[0], conv_w.shape[0]
    K = conv_w.shape[1] * patch * patch
    kpad = (K + 63) // 64 * 64
    G2 = (images.shape[2] // patch) * (images.shape[3] // patch)
    return [conv_w.new_empty((B * (G2 + 1), w)), conv_w.new_empty((B * G2, kpad), dtype=bf), conv_w.new_empty((B * (G2 + 1), w))]


def _vision_embed_bwd(dx0: Tensor, cols: Tensor, asm: Tensor, ln_w: Tensor, B: int, K: int, eps: float) -> List[Tensor]:
    """-> [dconv [w, K], dcls [w], dpos [S, w], dgamma, dbeta]"""
    w = asm.shape[1]
    S = asm.shape[0] // B
    d_asm, dg, db = ops.layernorm_bwd(asm, ln_w, dx0, eps)
    dpos = ops.colsum(d_asm.view(B, S * w)).view(S, w)  # asm[b, s] = (...) + pos[s]
    dcls = dpos[0].clone()                               # asm[b, 0] = cls + pos[0]: the same sum over the batch
    idx = (torch.arange(B * S, device=dx0.device, dtype=torch.int32).view(B, S)[:, 1:]).reshape(-1).contiguous()
    d_pe = ops.gather_rows(d_asm, w, idx, w, bf)        # rows of the patch tokens, bf16 [B*G2, w]
    dwk = _wgrad(d_pe, cols)                            # [w, kpad]
    dconv = dwk[:, :K].contiguous() if K != dwk.shape[1] else dwk
    return [dconv, dcls, dpos, dg, db]


def _vision_embed_bwd_fake(dx0, cols, asm, ln_w, B, K, eps):
    w = asm.shape[1]
    S = asm.shape[0] // B
    return [asm.new_empty((w, K)), asm.new_empty((w,)), asm.new_empty((S, w)), asm.new_empty((w,)), asm.new_empty((w,))]


vision_embed_fwd_op = _define("clip_vision_embed_fwd", "(Tensor images, Tensor conv_w, Tensor cls, Tensor pos, Tensor ln_w, Tensor ln_b, int patch, "
                              "float eps) -> Tensor[]", _vision_embed_fwd, _vision_embed_fwd_fake)
vision_embed_bwd_op = _define("clip_vision_embed_bwd", "(Tensor dx0, Tensor cols, Tensor asm, Tensor ln_w, int B, int K, float eps) -> Tensor[]",
                              _vision_embed_bwd, _vision_embed_bwd_fake)


class VisionEmbedFn(torch.autograd.Function):
    """images -> fp32 residual stream [B*(G2+1), w]: conv patch embedding, CLS, + positional embedding, ln_pre."""

    @staticmethod
    def forward(ctx, images, conv_w, cls, pos, ln_w, ln_b, patch: int, eps: float):
        x0, cols, asm = vision_embed_fwd_op(images.detach(), _c32(conv_w), _c32(cls), _c32(pos), _c32(ln_w), _c32(ln_b), patch, eps)
        ctx.save_for_backward(cols, asm, ln_w)
        ctx.meta = (images.shape[0], conv_w.shape[1] * patch * patch, tuple(conv_w.shape), eps, tuple(cls.shape), tuple(pos.shape))
        return x0

    @staticmethod
    def backward(ctx, dx0):
        cols, asm, ln_w = ctx.saved_tensors
        B, K, conv_shape, eps, cls_shape, pos_shape = ctx.meta
        dconv, dcls, dpos, dg, db = vision_embed_bwd_op(dx0.contiguous(), cols, asm, _c32(ln_w), B, K, eps)
        return None, dconv.view(conv_shape), dcls.view(cls_shape), dpos.view(pos_shape), dg, db, None, None


def _text_embed_bwd(dx0: Tensor, ids: Tensor, vocab: int, npos: int) -> List[Tensor]:
    """-> [dtable [vocab, d], dpos [npos, d]] for x0 = table[ids] + pos[:S]"""
    B, S = ids.shape
    d = dx0.shape[1]
    dpos = ops.colsum(dx0.view(B, S * d)).view(S, d)
    dtable = torch.zeros((vocab, d), dtype=f32, device=dx0.device)  # memset; rows collide -> fp32 atomics
    ops.scatter_add_rows_(dtable, ids.reshape(-1).contiguous(), dx0)
    dpos_full = dpos if npos == S else torch.cat([dpos, torch.zeros((npos - S, d), dtype=f32, device=dx0.device)])
    return [dtable, dpos_full]


text_embed_fwd_op = _define("clip_text_embed_fwd", "(Tensor ids, Tensor table, Tensor pos) -> Tensor", lambda ids, table, pos: ops.embed_tokens(ids, table, pos),
                            lambda ids, table, pos: table.new_empty((ids.shape[0] * ids.shape[1], table.shape[1])))
text_embed_bwd_op = _define("clip_text_embed_bwd", "(Tensor dx0, Tensor ids, int vocab, int npos) -> Tensor[]", _text_embed_bwd,
                            lambda dx0, ids, vocab, npos: [dx0.new_empty((vocab, dx0.shape[1])), dx0.new_empty((npos, dx0.shape[1]))])


class TextEmbedFn(torch.autograd.Function):
    """ids -> fp32 residual stream: token_embedding[ids] + positional_embedding."""

    @staticmethod
    def forward(ctx, ids, table, pos):
        x0 = text_embed_fwd_op(ids, _c32(table), _c32(pos))
        ctx.save_for_backward(ids)
        ctx.meta = (tuple(table.shape), tuple(pos.shape))
        return x0

    @staticmethod
    def backward(ctx, dx0):
        (ids,) = ctx.saved_tensors
        tshape, pshape = ctx.meta
        dtable, dpos = text_embed_bwd_op(dx0.contiguous(), ids, tshape[0], pshape[0])
        return None, dtable, dpos.view(pshape)


def _pooled_head_fwd(x: Tensor, idx64: Tensor, ln_w: Tensor, ln_b: Tensor, proj: Tensor, eps: float, proj_is_linear: bool) -> List[Tensor]:
    """-> [e [B, E], rows (the pooled rows of x), n (their LayerNorm)]"""
    B = idx64.numel()
    d = x.shape[1]
    rows = ops.gather_rows(x, d, idx64.to(torch.int32), d, f32)
    n = ops.layernorm(rows, ln_w, ln_b, eps, out_dtype=f32)
    E = proj.shape[0] if proj_is_linear else proj.shape[1]
    e = ops.f32_gemm_strided(n, d, 1, proj, d if proj_is_linear else 1, 1 if proj_is_linear else E, B, E, d)
    return [e, rows, n]


def _pooled_head_fwd_fake(x, idx64, ln_w, ln_b, proj, eps, proj_is_linear):
    B, d = idx64.numel(), x.shape[1]
    E = proj.shape[0] if proj_is_linear else proj.shape[1]
    return [x.new_empty((B, E)), x.new_empty((B, d)), x.new_empty((B, d))]


def _pooled_head_bwd(de: Tensor, rows: Tensor, n: Tensor, idx64: Tensor, ln_w: Tensor, proj: Tensor, eps: float, lin: bool, M: int) -> List[Tensor]:
    """-> [dx [M, d] (zero but for the pooled rows), dgamma, dbeta, dP]"""
    B, d = rows.shape
    E = proj.shape[0] if lin else proj.shape[1]
    if lin:   # P [E, d]: dP[j, k] = sum_b de[b, j] n[b, k];  dn[b, k] = sum_j de[b, j] P[j, k]
        dP = ops.f32_gemm_strided(de, 1, E, n, 1, d, E, d, B)
        dn = ops.f32_gemm_strided(de, E, 1, proj, 1, d, B, d, E)
    else:     # P [d, E]: dP[k, j] = sum_b n[b, k] de[b, j];  dn[b, k] = sum_j de[b, j] P[k, j]
        dP = ops.f32_gemm_strided(n, 1, d, de, 1, E, d, E, B)
        dn = ops.f32_gemm_strided(de, E, 1, proj, E, 1, B, d, E)
    drows, dg, db = ops.layernorm_bwd(rows, ln_w, dn, eps)
    dx = torch.zeros((M, d), dtype=f32, device=de.device)  # memset: only the pooled rows receive gradient
    ops.scatter_add_rows_(dx, idx64, drows)
    return [dx, dg, db, dP]


def _pooled_head_bwd_fake(de, rows, n, idx64, ln_w, proj, eps, lin, M):
    d = rows.shape[1]
    return [rows.new_empty((M, d)), rows.new_empty((d,)), rows.new_empty((d,)), torch.empty_like(proj)]


pooled_head_fwd_op = _define("clip_pooled_head_fwd", "(Tensor x, Tensor idx64, Tensor ln_w, Tensor ln_b, Tensor proj, float eps, bool proj_is_linear) -> Tensor[]",
                             _pooled_head_fwd, _pooled_head_fwd_fake)
pooled_head_bwd_op = _define("clip_pooled_head_bwd", "(Tensor de, Tensor rows, Tensor n, Tensor idx64, Tensor ln_w, Tensor proj, float eps, bool lin, int M) "
                             "-> Tensor[]", _pooled_head_bwd, _pooled_head_bwd_fake)


class PooledHeadFn(torch.autograd.Function):
    """x_L fp32 [B*S, d], row index per sample -> LayerNorm(row) @ P   (ln_post + projection / ln_final + EOT + projection)."""

    @staticmethod
    def forward(ctx, x, idx64, ln_w, ln_b, proj, eps: float, proj_is_linear: bool):
        xc = x.detach()
        e, rows, n = pooled_head_fwd_op(xc if xc.is_contiguous() else xc.contiguous(), idx64, _c32(ln_w), _c32(ln_b), _c32(proj), eps, proj_is_linear)
        ctx.save_for_backward(rows, n, idx64, ln_w, proj)
        ctx.meta = (eps, proj_is_linear, x.shape[0])
        return e

    @staticmethod
    def backward(ctx, de):
        rows, n, idx64, ln_w, proj = ctx.saved_tensors
        eps, lin, M = ctx.meta
        dx, dg, db, dP = pooled_head_bwd_op(de.contiguous(), rows, n, idx64, _c32(ln_w), _c32(proj), eps, lin, M)
        return dx, None, dg, db, dP, None, None
