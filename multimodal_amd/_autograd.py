"""Autograd nodes shared by the trainable model families: forward AND backward are HIP kernels behind
the C-ABI; torch.autograd only sequences them.

    EncoderStackFn   N pre-norm transformer layers (CLIP's torch-style layers, FLAVA's query/key/value layers via an adapter)
    LayerNormFn      LayerNorm over all rows of a [.., d] tensor
    RowsLinearFn     Linear (+bias) applied to one selected row per sample (CLS projections)
    L2NormalizeFn    F.normalize(dim=-1)

GEMM gradients: dgrad dX = dY W = gemm_bf16(dY, W^T) with the activation's backward fused into the epilogue where there is one;
wgrad dW = dY^T X = split-K gemm_bf16(dY^T, X^T) over the token index (operands transposed by a kernel that also yields the bias
gradient).  The residual-stream gradient stays fp32; MFMA operands are bf16.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import ops

bf, f32 = torch.bfloat16, torch.float32


def c32(t: Tensor) -> Tensor:
    t = t.detach()
    if t.dtype != f32:
        raise ops.MmamdError("training on the MI355X path keeps parameters in float32")
    return t if t.is_contiguous() else t.contiguous()


def dgrad(dy: Tensor, w: Tensor, out_dtype, act: int = ops.ACT_NONE, pre_act: Optional[Tensor] = None) -> Tensor:
    """dX[M,K] = dY[M,N] . W[N,K]  (W fp32 [N,K], N % 64 == 0); with act = ACT_MUL_*_GRAD the epilogue multiplies by
    act'(pre_act): the activation's backward without a pass of its own."""
    wT = ops.transpose_to_bf16(w, pad_to=64)  # bf16 [K, N rounded up to 64, zero tail]
    if wT.shape[1] != dy.shape[1]:            # e.g. a 96-wide vocabulary: pad the contraction with zero columns
        pad = torch.zeros((dy.shape[0], wT.shape[1]), dtype=bf, device=dy.device)
        pad[:, :dy.shape[1]].copy_(dy)
        dy = pad
    return ops.gemm_bf16(dy, wT, None, act=act, residual=pre_act, out_dtype=out_dtype)


def dgrad_t(dy: Tensor, wT: Tensor, out_dtype, act: int = ops.ACT_NONE, pre_act: Optional[Tensor] = None) -> Tensor:
    """dgrad with the bf16 transpose of W already made (ops.pack_weights: one launch per stack and step instead of one transpose per GEMM)."""
    if wT.shape[1] != dy.shape[1]:
        pad = torch.zeros((dy.shape[0], wT.shape[1]), dtype=bf, device=dy.device)
        pad[:, :dy.shape[1]].copy_(dy)
        dy = pad
    return ops.gemm_bf16(dy, wT, None, act=act, residual=pre_act, out_dtype=out_dtype)


_FUSED_BIAS_GRAD = True  # tools/train_bench.py --no-fused-bias flips it for the A/B
_BF16_DH = True  # the two dgrad GEMMs in front of a LayerNorm backward hand it bf16 (the residual-stream gradient stays fp32): -0.57 ms on the CLIP step (profiles/r05_train_bf16_dh_ab2.txt); tools/train_bench.py --f32-dh is the other arm
_DEFER_LN_REDUCE = True  # tools/train_bench.py --no-deferred-ln-reduce: every LayerNorm backward reduces its own partials (the r04 form)


def _pending():
    """The list the LayerNorm backward calls of a stack park their reductions in (None: reduce at once)."""
    return [] if _DEFER_LN_REDUCE else None


def wgrad(dy: Tensor, x: Tensor, bias: bool = False):
    """dW[N,K] = dY^T X for dy [M,N], x [M,K] (bf16 or fp32 row-major); contraction over the M tokens.  bias=True also returns
    db[N] = column sums of the bf16-rounded dY.  Token counts that are multiples of 128 (every full-size batch) go straight from the
    row-major operands (mmamd_gemm_bf16_tn_splitk); others take transposed, zero-padded copies (db then comes out of the transpose pass)."""
    if dy.shape[0] % 128 == 0 and dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0:
        dyb = dy if dy.dtype == bf else ops.convert(dy, bf)
        xb = x if x.dtype == bf else ops.convert(x, bf)
        if bias and _FUSED_BIAS_GRAD:  # db from the wgrad GEMM's own pass over dY (no column-sum pass)
            return ops.gemm_bf16_tn_splitk(dyb, xb, want_colsum=True)
        dW = ops.gemm_bf16_tn_splitk(dyb, xb)
        return (dW, ops.colsum(dyb)) if bias else dW
    if bias:
        dyT, db = ops.transpose_to_bf16(dy, with_colsum=True)
    else:
        dyT, db = ops.transpose_to_bf16(dy), None
    dW = ops.gemm_bf16_splitk(dyT, ops.transpose_to_bf16(x))
    return (dW, db) if bias else dW


_GROUPED_WGRAD = True  # tools/train_bench.py --no-grouped-wgrad: one split-K launch (+ reduce) per Linear, the form before r05's grouped launch


def wgrad_many(jobs):
    """[(dy, x, bias)] -> [(dW, db or None)]: the weight / bias gradients of a layer's Linears.  Full-size batches (token counts that are multiples of 128)
    go to ONE grouped split-K launch + one reduce launch (mmamd_gemm_bf16_tn_splitk_group: together the four problems of a layer fill the CUs at 5-7
    splits each, where the small ones alone need 21-28 and every one its own reduce); everything else takes wgrad() one by one."""
    ok = _GROUPED_WGRAD and _FUSED_BIAS_GRAD and 1 < len(jobs) <= 8 and all(
        dy.shape[0] % 128 == 0 and dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0 and dy.shape[0] == x.shape[0] for dy, x, _ in jobs)
    if not ok:
        return [wgrad(dy, x, bias=True) if b else (wgrad(dy, x), None) for dy, x, b in jobs]
    return ops.gemm_bf16_tn_splitk_group([(dy if dy.dtype == bf else ops.convert(dy, bf), x if x.dtype == bf else ops.convert(x, bf), b) for dy, x, b in jobs])


class StackConfig:
    """Static description of a layer stack for EncoderStackFn.  to_canonical(layer params) -> the 12 canonical tensors
    (Wqkv [3d,d], bqkv, Wo, bo, W1, b1, W2, b2, g1, be1, g2, be2); from_canonical(12 grads) -> grads in the layer's parameter order."""

    def __init__(self, n_layers: int, n_head: int, B: int, S: int, causal: bool, act: int, eps1: Sequence[float], eps2: Sequence[float],
                 params_per_layer: int, to_canonical: Callable, from_canonical: Callable, key_mask: Optional[Tensor] = None,
                 keep_hidden: bool = False, drop: Optional[Sequence[float]] = None, seed: int = 0, norm_first: bool = True,
                 full_mask: Optional[Tensor] = None, head_mask: Optional[Tensor] = None):
        self.n_layers, self.n_head, self.B, self.S, self.causal, self.act = n_layers, n_head, B, S, causal, act
        self.full_mask = full_mask  # uint8 [B or 1, S, S] (0 = masked): arbitrary attention masks go through the general attention kernels
        # the reference's head_mask (modules/layers/attention.py:236-237): fp32, broadcastable to [B, H, S, S], multiplied into the probabilities after
        # softmax (and dropout) in every layer of the stack; a constant (no gradient); general attention kernels, forward and backward
        self.head_mask = head_mask
        self.norm_first = bool(norm_first)  # False: the reference's DEFAULT post-norm layers (modules/layers/transformer.py:56,118-132)
        # training-time dropout (stack_drop_spec): [] = none, else [p_branch, p_mlp, p_attn] + one stochastic-depth rate per layer (-1 = none)
        self.drop, self.seed = [float(v) for v in (drop or [])], int(seed)
        self.eps1, self.eps2, self.ppl = list(eps1), list(eps2), params_per_layer
        self.to_canonical, self.from_canonical, self.key_mask = to_canonical, from_canonical, key_mask
        self.keep_hidden = keep_hidden  # the node then returns (x_L, inputs of layers 1 .. N-1) and fills `qkv`
        self.qkv: List[Tensor] = []
        self.lse: List[Tensor] = []  # ... and their log2-domain log-sum-exp rows [B, H, S] (probabilities without a second attention pass)


def draw_seed() -> int:
    """A fresh 62-bit Philox key for the dropout masks of ONE forward, drawn on the host from torch's CPU generator (reproducible under
    torch.manual_seed); the backward regenerates its masks from the same key, nothing is stored.  Not under graph capture / torch.compile:
    the key would be baked into the graph and every replay would drop the same elements."""
    if torch.compiler.is_compiling():
        raise ops.MmamdError("training-time dropout on the MI355X path is an eager-mode feature (torch.compile would bake the mask key into the graph)")
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise ops.MmamdError("training-time dropout cannot be captured in a HIP graph (the mask key is drawn on the host per step)")
    return int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())


def branch_rates(mod) -> Tuple[float, float]:
    """(dropout p, stochastic-depth rate or -1) of a residual-branch module: nn.Dropout or StochasticDepth (modules/layers/stochastic_depth.py)."""
    from .modules.layers.stochastic_depth import StochasticDepth

    if isinstance(mod, StochasticDepth):
        if mod.mode != "row":
            raise ops.MmamdError("stochastic depth inside a layer stack: mode='row' only (what the reference's layers use)")
        return 0.0, float(mod.p)
    return float(getattr(mod, "p", 0.0)), -1.0


def stack_drop_spec(layers, attn_p: Callable = None, training: bool = True) -> Tuple[List[float], int]:
    """Dropout description of a stack of pre-norm layers whose members follow the reference's naming (attention_dropout, feedforward_dropout,
    feedforward = MLP): ([], 0) when every rate is zero, else ([p_branch, p_mlp, p_attn] + per-layer stochastic-depth rates, fresh seed).
    One p_branch / p_mlp / p_attn per stack (the reference builds all layers of a stack from one `dropout` value; transformer.py:175-200).
    training=False (an eval-mode stack on the differentiable path because its INPUT requires grad): ([], 0) -- nn.Dropout and StochasticDepth
    are the identity in eval mode whatever their rates."""
    if not training:
        return [], 0
    pb, pm, pa, path = set(), set(), set(), []
    for layer in layers:
        p1, r1 = branch_rates(layer.attention_dropout)
        p2, r2 = branch_rates(layer.feedforward_dropout)
        if (p1, r1) != (p2, r2):
            raise ops.MmamdError("training: attention_dropout and feedforward_dropout of a layer must have the same rate")
        pb.add(p1)
        path.append(r1)
        pm.add(layer.feedforward.hidden_dropout_p())
        pa.add(float(attn_p(layer)) if attn_p is not None else 0.0)
    if len(pb) > 1 or len(pm) > 1 or len(pa) > 1:
        raise ops.MmamdError("training: all layers of a stack must share their dropout rates")
    p_branch, p_mlp, p_attn = pb.pop(), pm.pop(), pa.pop()
    if p_branch == 0 and p_mlp == 0 and p_attn == 0 and all(r <= 0 for r in path):
        return [], 0
    if all(r < 0 for r in path):
        path = []
    return [p_branch, p_mlp, p_attn] + [max(r, -1.0) for r in path], draw_seed()


class DropoutFn(torch.autograd.Function):
    """y = x * keep / (1 - p) (mmamd_dropout; group > 0: one decision per sample of `group` elements = stochastic depth, mode 'row').  The
    backward is the same kernel on the incoming gradient: the mask is a function of (seed, site, index)."""

    @staticmethod
    def forward(ctx, x, p: float, seed: int, site: int, group: int):
        xc = x.detach()
        xc = xc if xc.is_contiguous() else xc.contiguous()
        ctx.meta = (p, seed, site, group)
        return ops.dropout(xc, p, seed, site, group=group)

    @staticmethod
    def backward(ctx, dy):
        p, seed, site, group = ctx.meta
        d = dy.detach()
        return ops.dropout(d if d.is_contiguous() else d.contiguous(), p, seed, site, group=group), None, None, None, None


def dropout_train(x: Tensor, p: float, site: int = 0, group: int = 0) -> Tensor:
    """nn.Dropout(p) / StochasticDepth(p) of a TRAINING forward on a contiguous fp32 / bf16 tensor whose element count is a multiple of 4."""
    if p <= 0:
        return x
    if x.numel() % 4 != 0:
        raise ops.MmamdError(f"dropout on the MI355X path needs a multiple of 4 elements, got {tuple(x.shape)}")
    return DropoutFn.apply(x, float(p), draw_seed(), site, group)


_PACK_WEIGHTS = True  # tools/train_pack_ab.py flips it for the A/B


def _drop_of(drop: List[float], li: int, S: int, d: int) -> Tuple[float, float, int]:
    """(p of the two residual branches of layer li, p of the MLP's hidden dropout, elements per sample or 0): a stochastic-depth rate
    replaces the branch dropout of its layer (reference transformer.py:64-70)."""
    if not drop:
        return 0.0, 0.0, 0
    path = drop[3 + li] if len(drop) > 3 else -1.0
    return (path, drop[1], S * d) if path >= 0 else (drop[0], drop[1], 0)


def _saved_per_layer(norm_first: bool) -> int:
    return 8 if norm_first else 9


def _stack_fwd_impl(x0: Tensor, params: List[Tensor], n_head: int, B: int, S: int, causal: bool, act: int, eps1: List[float],
                    eps2: List[float], key_mask: Optional[Tensor], drop: List[float], seed: int, norm_first: bool = True,
                    full_mask: Optional[Tensor] = None, head_mask: Optional[Tensor] = None) -> List[Tensor]:
    """Forward of N layers.  params: the 12 canonical fp32 tensors per layer.  Returns [x_L] + per layer
    [h1, qkv, att, lse, x_mid, h2, u, g] (+ [ff] for post-norm layers) + the inputs of layers 1 .. N-1 (layer 0's input is x0 itself).
    Pre-norm (reference transformer.py:96-116):  x_mid = x + attn(LN1 x);  x' = x_mid + mlp(LN2 x_mid)
    Post-norm (:118-132, the reference's default): a = x + attn(x);  x1 = LN1 a;  ff = x1 + mlp(x1);  x' = LN2 ff
      -- saved under the same names: h1 = bf16(x), x_mid = a, h2 = bf16(x1), and ff (the input of LN2) as a ninth tensor."""
    H = n_head
    n_layers = len(params) // 12
    saved: List[Tensor] = []
    inputs: List[Tensor] = []
    # every Linear weight of the stack -> bf16 (forward operand) and bf16 transpose (dgrad operand, kept for the backward): one launch per 64
    ws = [params[12 * li + k] for li in range(n_layers) for k in (0, 2, 4, 6)]
    if _PACK_WEIGHTS:
        wb, wt = ops.pack_weights(ws)
    else:  # (A/B: one convert and one transpose launch per weight, as before r03)
        wb, wt = [ops.convert(w, bf) for w in ws], [ops.transpose_to_bf16(w, pad_to=64) for w in ws]
    x = x0
    for li in range(n_layers):
        _, bqkv, _, bo, _, b1, _, b2, g1, be1, g2, be2 = params[12 * li:12 * li + 12]
        Wqkv, Wo, W1, W2 = wb[4 * li:4 * li + 4]
        if li > 0:
            inputs.append(x)
        h1 = ops.layernorm(x, g1, be1, eps1[li], out_dtype=bf) if norm_first else ops.convert(x, bf)
        qkv = ops.gemm_bf16(h1, Wqkv, bqkv)
        att, lse = _attn_fwd_any(qkv, B, S, H, causal, key_mask, full_mask, drop[2] if drop else 0.0, seed, 16 * li + 3, head_mask)
        pb, pm, grp = _drop_of(drop, li, S, x.shape[1])
        if pb > 0:  # x_mid = x + drop(att Wo^T + bo): the projection without the residual, then ONE pass: mask, scale, add
            x_mid = ops.dropout(ops.gemm_bf16(att, Wo, bo, out_dtype=f32), pb, seed, 16 * li, residual=x, group=grp)
        else:
            x_mid = ops.gemm_bf16(att, Wo, bo, residual=x, out_dtype=f32, out=torch.empty_like(x))
        if norm_first:
            res2 = x_mid
            h2 = ops.layernorm(x_mid, g2, be2, eps2[li], out_dtype=bf)
        else:  # post-norm: the MLP reads (and adds onto) the NORMALISED sum
            res2 = ops.layernorm(x_mid, g1, be1, eps1[li], out_dtype=f32)
            h2 = ops.convert(res2, bf)
        u, g = ops.gemm_bf16_dual(h2, W1, b1, act)  # pre-activation (kept for the backward) + activation
        if pm > 0:
            ops.dropout(g, pm, seed, 16 * li + 1, out=g)  # the MLP's hidden dropout, in place: the dropped g feeds W2 and its gradient
        if pb > 0:
            x_out = ops.dropout(ops.gemm_bf16(g, W2, b2, out_dtype=f32), pb, seed, 16 * li + 2, residual=res2, group=grp)
        else:
            x_out = ops.gemm_bf16(g, W2, b2, residual=res2, out_dtype=f32, out=torch.empty_like(x))
        saved += [h1, qkv, att, lse, x_mid, h2, u, g]
        if not norm_first:
            saved.append(x_out)  # ff: the input of LN2
            x_out = ops.layernorm(x_out, g2, be2, eps2[li], out_dtype=f32)
        x = x_out
    if n_layers == 0:
        x = x0.clone()  # a custom op (mutates_args = ()) must not return an alias of its input
    return [x] + saved + inputs + wt


def _stack_fwd_fake(x0, params, n_head, B, S, causal, act, eps1, eps2, key_mask, drop, seed, norm_first=True, full_mask=None, head_mask=None):
    n_layers = len(params) // 12
    M, d = x0.shape
    saved, inputs = [], []
    for li in range(n_layers):
        ff = params[12 * li + 4].shape[0]
        if li > 0:
            inputs.append(x0.new_empty((M, d)))
        saved += [x0.new_empty((M, d), dtype=bf), x0.new_empty((M, 3 * d), dtype=bf), x0.new_empty((M, d), dtype=bf),
                  x0.new_empty((B, n_head, S)), x0.new_empty((M, d)), x0.new_empty((M, d), dtype=bf), x0.new_empty((M, ff), dtype=bf),
                  x0.new_empty((M, ff), dtype=bf)]
        if not norm_first:
            saved.append(x0.new_empty((M, d)))
    wt = []
    for li in range(n_layers):
        for k in (0, 2, 4, 6):
            N, K = params[12 * li + k].shape
            wt.append(x0.new_empty((K, (N + 63) // 64 * 64), dtype=bf))
    return [x0.new_empty((M, d))] + saved + inputs + wt


_ACT_GRAD = {ops.ACT_QUICKGELU: ops.ACT_MUL_QUICKGELU_GRAD, ops.ACT_GELU_ERF: ops.ACT_MUL_GELU_GRAD}


def _stack_bwd_impl(dx_out: Tensor, x0: Tensor, saved: List[Tensor], params: List[Tensor], n_head: int, B: int, S: int, causal: bool,
                    act: int, eps1: List[float], eps2: List[float], key_mask: Optional[Tensor], drop: List[float], seed: int,
                    dhidden: List[Optional[Tensor]], norm_first: bool = True, full_mask: Optional[Tensor] = None,
                    head_mask: Optional[Tensor] = None) -> List[Tensor]:
    """Backward of _stack_fwd_impl: saved = its outputs [1:].  Returns [dX0] + the 12 canonical gradients per layer.  dhidden (empty, or one
    entry per layer 1 .. N-1): gradients that arrived through the intermediate hidden states the forward handed out (None = unused)."""
    H = n_head
    n_layers = len(params) // 12
    ns = _saved_per_layer(norm_first)
    if not norm_first:
        return _stack_bwd_postnorm(dx_out, x0, saved, params, n_head, B, S, causal, act, eps1, eps2, key_mask, drop, seed, dhidden, full_mask, head_mask)
    inputs = [x0] + list(saved[ns * n_layers:(ns + 1) * n_layers - 1])
    wt = saved[(ns + 1) * n_layers - 1:]  # bf16 transposes of (Wqkv, Wo, W1, W2) per layer, made by the forward's weight pack
    dX = dx_out
    grads: List[Tensor] = [dX] * (12 * n_layers)
    pending = _pending()  # the LayerNorm backward calls park their dgamma / dbeta / column-sum partials: ONE reduction launch at the end of the stack
    dXb = None  # bf16 copy of dX: produced by the LayerNorm backward of the layer above
    dXsum = None  # ... and its column sums (= the bias gradient of this layer's second MLP Linear) from the same kernel
    for li in reversed(range(n_layers)):
        h1, qkv, att, lse, x_mid, h2, u, g = saved[8 * li:8 * li + 8]
        x = inputs[li]
        Wqkv, bqkv, Wo, bo, W1, b1, W2, b2, g1, be1, g2, be2 = params[12 * li:12 * li + 12]
        pb, pm, grp = _drop_of(drop, li, S, x.shape[1])
        if pb > 0:  # x_out = x_mid + drop(delta): the branch gradient is the masked, scaled dX (bf16 for the GEMMs); x_mid's share stays dX
            dXb, dXsum = ops.dropout(dX, pb, seed, 16 * li + 2, group=grp, out_dtype=bf), None
        elif dXb is None:
            dXb = ops.convert(dX, bf)
        # x_out = x_mid + g W2^T + b2;  g = act(u): du = (dX W2) * act'(u) in the dgrad GEMM's epilogue
        WqkvT, WoT, W1T, W2T = wt[4 * li:4 * li + 4]
        du = dgrad_t(dXb, W2T, bf, _ACT_GRAD[act], u)
        if pm > 0:
            ops.dropout(du, pm, seed, 16 * li + 1, out=du)  # g' = g * m / (1 - p): the mask commutes with the activation's derivative
        wjobs = [(dXb, g, dXsum is None)]  # the layer's four weight gradients go out together below (nothing on the way down needs them)
        # u = h2 W1^T + b1
        dh2 = dgrad_t(du, W1T, bf if _BF16_DH else f32)
        wjobs.append((du, h2, True))
        dx_mid, dg2, dbe2, dxmb, dbo = ops.layernorm_bwd(x_mid, g2, dh2, eps2[li], add=dX, want_bf16=True, want_colsum=True, defer=pending)
        if pb > 0:  # x_mid = x + drop(att Wo^T + bo)
            dxmb = ops.dropout(dx_mid, pb, seed, 16 * li, group=grp, out_dtype=bf)
            dbo = ops.colsum(dxmb)
        # x_mid = x + att Wo^T + bo
        datt = dgrad_t(dxmb, WoT, bf)
        wjobs.append((dxmb, att, False))
        dqkv = _attn_bwd_any(qkv, att, datt, lse, B, S, H, causal, key_mask, full_mask, drop[2] if drop else 0.0, seed, 16 * li + 3, head_mask)
        # qkv = h1 Wqkv^T + bqkv
        dh1 = dgrad_t(dqkv, WqkvT, bf if _BF16_DH else f32)
        wjobs.append((dqkv, h1, True))
        (dW2, db2), (dW1, db1), (dWo, _), (dWqkv, dbqkv) = wgrad_many(wjobs)
        if dXsum is not None:
            db2 = dXsum
        dX, dg1, dbe1, dXb, dXsum = ops.layernorm_bwd(x, g1, dh1, eps1[li], add=dx_mid, want_bf16=True, want_colsum=True, defer=pending)
        if li > 0 and dhidden and dhidden[li - 1] is not None:
            # this layer's input was also handed out as hidden_states[li] and something differentiated through it: dX += that gradient
            # (mmamd_dropout with p = 0 is the fp32 add kernel); the bf16 copy / column sums of dX made above are stale
            extra = dhidden[li - 1].detach()
            dX = ops.dropout(extra if extra.is_contiguous() else extra.contiguous(), 0.0, 0, 0, residual=dX)
            dXb, dXsum = None, None
        grads[12 * li:12 * li + 12] = [dWqkv, dbqkv, dWo, dbo, dW1, db1, dW2, db2, dg1, dbe1, dg2, dbe2]
    ops.colsum_flush(pending)
    if n_layers == 0:
        dX = dx_out.clone()  # (no alias of an input, see _stack_fwd_impl)
    return [dX] + grads


def _attn_fwd_any(qkv, B, S, H, causal, key_mask, full_mask, pa, seed, site, head_mask=None):
    """(att, lse) of the self-attention of a stack layer: the packed-qkv flash kernels, or the general kernels when the probabilities carry
    dropout (FLAVA's SelfAttention(dropout): the kernel generates the Philox mask) or the caller gave an arbitrary [Sq, Sk] mask."""
    if pa > 0 or full_mask is not None or head_mask is not None:
        dm = qkv.shape[1] // 3
        lse = torch.empty((B, H, S), dtype=f32, device=qkv.device)
        att, _ = ops.attention_x_fwd(qkv[:, :dm], qkv[:, dm:2 * dm], qkv[:, 2 * dm:], B, S, S, H, dm // H,
                                     ops.AttnMask(causal=causal, key_mask=key_mask, full=full_mask), lse=lse,
                                     drop=(pa, seed, site) if pa > 0 else None, head_mask=head_mask)
        return att, lse
    return ops.attention_fwd_train(qkv, B, S, H, causal, key_mask)


def _attn_bwd_any(qkv, att, datt, lse, B, S, H, causal, key_mask, full_mask, pa, seed, site, head_mask=None):
    """dqkv of _attn_fwd_any."""
    if pa > 0 or full_mask is not None or head_mask is not None:
        dm = att.shape[1]
        dq_, dkv_ = ops.attention_x_bwd(qkv[:, :dm], qkv[:, dm:2 * dm], qkv[:, 2 * dm:], att, datt, lse, B, S, S, H, dm // H,
                                        ops.AttnMask(causal=causal, key_mask=key_mask, full=full_mask),
                                        drop=(pa, seed, site) if pa > 0 else None, head_mask=head_mask)
        dqkv = torch.empty((B * S, 3 * dm), dtype=bf, device=att.device)  # [dq | dk | dv]: placement copies of the two kernel outputs
        dqkv[:, :dm].copy_(dq_)
        dqkv[:, dm:].copy_(dkv_)
        return dqkv
    return ops.attention_bwd(qkv, att, datt, lse, B, S, H, causal, key_mask)


def _stack_bwd_postnorm(dx_out, x0, saved, params, n_head, B, S, causal, act, eps1, eps2, key_mask, drop, seed, dhidden, full_mask=None,
                        head_mask=None) -> List[Tensor]:
    """Backward of the post-norm layers (reference transformer.py:118-132): y = LN2(ff), ff = x1 + mlp(x1), x1 = LN1(a), a = x + attn(x).
    The same kernels as the pre-norm backward in another order: each LayerNorm backward now sits at the END of its block, and the two
    residual sums are epilogues of the dgrad GEMMs (dx1 = dff + du W1, dx = da + dqkv Wqkv)."""
    H = n_head
    n_layers = len(params) // 12
    wt = saved[10 * n_layers - 1:]
    dX = dx_out
    grads: List[Tensor] = [dX] * (12 * n_layers)
    pending = _pending()  # parked LayerNorm-backward reductions: one launch at the end (ops.colsum_flush)
    for li in reversed(range(n_layers)):
        h1, qkv, att, lse, a, h2, u, g, ff = saved[9 * li:9 * li + 9]
        Wqkv, bqkv, Wo, bo, W1, b1, W2, b2, g1, be1, g2, be2 = params[12 * li:12 * li + 12]
        WqkvT, WoT, W1T, W2T = wt[4 * li:4 * li + 4]
        pb, pm, grp = _drop_of(drop, li, S, a.shape[1])
        # y = LN2(ff)
        dff, dg2, dbe2, dffb, db2 = ops.layernorm_bwd(ff, g2, dX, eps2[li], want_bf16=True, want_colsum=True, defer=pending)
        if pb > 0:  # ff = x1 + drop(delta): the branch gradient is the masked, scaled dff
            dffb = ops.dropout(dff, pb, seed, 16 * li + 2, group=grp, out_dtype=bf)
            db2 = ops.colsum(dffb)
        du = dgrad_t(dffb, W2T, bf, _ACT_GRAD[act], u)
        if pm > 0:
            ops.dropout(du, pm, seed, 16 * li + 1, out=du)
        dW2 = wgrad(dffb, g)
        dx1 = dgrad_t(du, W1T, f32, ops.ACT_NONE, dff)  # dff + du W1: the residual add is the GEMM's epilogue
        dW1, db1 = wgrad(du, h2, bias=True)
        # x1 = LN1(a)
        da, dg1, dbe1, dab, dbo = ops.layernorm_bwd(a, g1, dx1, eps1[li], want_bf16=True, want_colsum=True, defer=pending)
        if pb > 0:
            dab = ops.dropout(da, pb, seed, 16 * li, group=grp, out_dtype=bf)
            dbo = ops.colsum(dab)
        datt = dgrad_t(dab, WoT, bf)
        dWo = wgrad(dab, att)
        dqkv = _attn_bwd_any(qkv, att, datt, lse, B, S, H, causal, key_mask, full_mask, drop[2] if drop else 0.0, seed, 16 * li + 3, head_mask)
        dX = dgrad_t(dqkv, WqkvT, f32, ops.ACT_NONE, da)  # da + dqkv Wqkv
        dWqkv, dbqkv = wgrad(dqkv, h1, bias=True)
        if li > 0 and dhidden and dhidden[li - 1] is not None:
            extra = dhidden[li - 1].detach()
            dX = ops.dropout(extra if extra.is_contiguous() else extra.contiguous(), 0.0, 0, 0, residual=dX)
        grads[12 * li:12 * li + 12] = [dWqkv, dbqkv, dWo, dbo, dW1, db1, dW2, db2, dg1, dbe1, dg2, dbe2]
    ops.colsum_flush(pending)
    if n_layers == 0:
        dX = dx_out.clone()
    return [dX] + grads


def _stack_bwd_fake(dx_out, x0, saved, params, n_head, B, S, causal, act, eps1, eps2, key_mask, drop, seed, dhidden, norm_first=True, full_mask=None,
                    head_mask=None):
    return [torch.empty_like(x0)] + [torch.empty_like(p) for p in params]


from ._custom_op import define as _define  # noqa: E402

_STACK_SCALARS = "int n_head, int B, int S, bool causal, int act, float[] eps1, float[] eps2, Tensor? key_mask, float[] drop, int seed"
stack_fwd_op = _define("encoder_stack_fwd", f"(Tensor x0, Tensor[] params, {_STACK_SCALARS}, bool norm_first=True, Tensor? full_mask=None, Tensor? head_mask=None) -> Tensor[]", _stack_fwd_impl, _stack_fwd_fake)
stack_bwd_op = _define("encoder_stack_bwd", f"(Tensor dx_out, Tensor x0, Tensor[] saved, Tensor[] params, {_STACK_SCALARS}, Tensor?[] dhidden, bool norm_first=True, Tensor? full_mask=None, Tensor? head_mask=None) -> Tensor[]",
                       _stack_bwd_impl, _stack_bwd_fake)


class EncoderStackFn(torch.autograd.Function):
    """x0 fp32 [B*S, d] -> x_L.  Per layer (tensors kept for backward in brackets):
        [x] -LN-> [h1] -GEMM-> [qkv] -attention-> [att, lse] -GEMM(+x)-> [x_mid] -LN-> [h2] -GEMM-> [u] -act-> [g] -GEMM(+x_mid)-> x'
    Forward and backward are ONE dispatcher op each (torch.ops.mmamd_train.encoder_stack_fwd / _bwd, multimodal_amd/_custom_op.py)."""

    @staticmethod
    def forward(ctx, x0: Tensor, cfg: StackConfig, *params: Tensor):
        x = x0.detach()
        x = x if x.is_contiguous() else x.contiguous()
        canon: List[Tensor] = []
        for li in range(cfg.n_layers):
            canon += list(cfg.to_canonical([c32(p) for p in params[cfg.ppl * li:cfg.ppl * (li + 1)]]))
        outs = stack_fwd_op(x, canon, cfg.n_head, cfg.B, cfg.S, cfg.causal, cfg.act, cfg.eps1, cfg.eps2, cfg.key_mask, cfg.drop, cfg.seed, cfg.norm_first, cfg.full_mask, cfg.head_mask)
        ns = _saved_per_layer(cfg.norm_first)
        ctx.save_for_backward(x, *outs[1:], *params)
        ctx.cfg, ctx.nparam = cfg, len(params)
        ctx.set_materialize_grads(False)
        if cfg.keep_hidden:
            # the inputs of layers 1 .. N-1 are OUTPUTS of this node too: hidden_states[1 .. N-1] of the reference's TransformerOutput stay
            # attached to the graph in training like the reference's (models/flava/transformer.py:254-259); hidden_states[0] is the caller's own
            # input tensor and hidden_states[N] the result.  cfg.qkv: the packed projections of every layer, for callers that also hand out
            # attention probabilities in training (they recompute them from here, detached)
            mids = list(outs[1 + ns * cfg.n_layers:(ns + 1) * cfg.n_layers])
            cfg.qkv = [outs[1 + ns * li + 1] for li in range(cfg.n_layers)]
            cfg.lse = [outs[1 + ns * li + 3] for li in range(cfg.n_layers)]
            ctx.n_mid = len(mids)
            return (outs[0], *mids)
        ctx.n_mid = -1
        return outs[0]

    @staticmethod
    def backward(ctx, dx_out, *dmid):
        cfg, nparam = ctx.cfg, ctx.nparam
        tensors = ctx.saved_tensors
        x0, saved, params = tensors[0], tensors[1:len(tensors) - nparam], tensors[len(tensors) - nparam:]
        if dx_out is None:  # only intermediate hidden states were differentiated
            dX = torch.zeros_like(x0)  # memset
        else:
            dX = dx_out.detach()
            dX = dX if dX.is_contiguous() else dX.contiguous()
        canon: List[Tensor] = []
        for li in range(cfg.n_layers):
            canon += list(cfg.to_canonical([c32(p) for p in params[cfg.ppl * li:cfg.ppl * (li + 1)]]))
        dhidden = list(dmid) if any(d is not None for d in dmid) else []
        outs = stack_bwd_op(dX, x0, list(saved), canon, cfg.n_head, cfg.B, cfg.S, cfg.causal, cfg.act, cfg.eps1, cfg.eps2, cfg.key_mask, cfg.drop,
                            cfg.seed, dhidden, cfg.norm_first, cfg.full_mask, cfg.head_mask)
        grads: List[Optional[Tensor]] = [None] * nparam
        for li in range(cfg.n_layers):
            grads[cfg.ppl * li:cfg.ppl * (li + 1)] = cfg.from_canonical(list(outs[1 + 12 * li:13 + 12 * li]))
        return (outs[0], None, *grads)


class LayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x) over the last dimension of a contiguous fp32 tensor (affine)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps: float):
        xc = x.detach()
        xc = xc if xc.is_contiguous() else xc.contiguous()
        if xc.dtype != f32:
            raise ops.MmamdError("differentiable LayerNorm on the MI355X path takes fp32 activations")
        y = ops.layernorm(xc, c32(weight), c32(bias), eps, out_dtype=f32)
        ctx.save_for_backward(xc, weight)
        ctx.eps = eps
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dyc = dy.detach()
        dyc = dyc if dyc.is_contiguous() else dyc.contiguous()
        dx, dg, db = ops.layernorm_bwd(x, c32(weight), dyc, ctx.eps)
        return dx.view(x.shape), dg, db, None


class RowsLinearFn(torch.autograd.Function):
    """out[b] = x2d[rows[b]] . W^T (+ bias): a Linear applied to one selected row per sample (exact-fp32 MFMA both ways)."""

    @staticmethod
    def forward(ctx, x2d, rows64, weight, bias):
        d = x2d.shape[1]
        xc = x2d.detach()
        xc = xc if xc.is_contiguous() else xc.contiguous()
        sel = ops.gather_rows(xc, d, rows64.to(torch.int32), d, f32)
        W = c32(weight)
        E = W.shape[0]
        out = ops.rows_linear_f32(sel, d, sel.shape[0], W, c32(bias) if bias is not None else None)
        ctx.save_for_backward(sel, rows64, weight)
        ctx.meta = (tuple(x2d.shape), E, bias is not None)
        return out

    @staticmethod
    def backward(ctx, de):
        sel, rows64, weight = ctx.saved_tensors
        xshape, E, has_bias = ctx.meta
        B, d = sel.shape
        de = de.detach()
        de = de if de.is_contiguous() else de.contiguous()
        W = c32(weight)
        dW = ops.f32_gemm_strided(de, 1, E, sel, 1, d, E, d, B)   # dW[j,k] = sum_b de[b,j] sel[b,k]
        dsel = ops.f32_gemm_strided(de, E, 1, W, 1, d, B, d, E)   # dsel[b,k] = sum_j de[b,j] W[j,k]
        db = ops.colsum(de) if has_bias else None
        dx = torch.zeros(xshape, dtype=f32, device=de.device)      # memset: only the selected rows receive gradient
        ops.scatter_add_rows_(dx, rows64, dsel)
        return dx, None, dW, db


class SmallLinearF32Fn(torch.autograd.Function):
    """y = act(x W^T + b) in exact fp32 for a FEW rows (classifier heads on the CLS row): x fp32 [M, K] contiguous; act in
    {None, "relu"}.  Forward mmamd_rows_linear_f32, backward two strided exact-fp32 GEMMs + a column sum (+ mmamd_relu_bwd)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu: bool):
        xc = c32(x)
        W = c32(weight)
        y = ops.rows_linear_f32(xc, xc.shape[1], xc.shape[0], W, c32(bias) if bias is not None else None, relu=relu)
        ctx.save_for_backward(xc, weight, y if relu else xc.new_empty(0))
        ctx.meta = (relu, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, weight, y = ctx.saved_tensors
        relu, has_bias = ctx.meta
        M, K = xc.shape
        W = c32(weight)
        E = W.shape[0]
        dz = dy.detach()
        dz = dz if dz.is_contiguous() else dz.contiguous()
        if relu:
            dz = ops.relu_bwd(y, dz)
        dW = ops.f32_gemm_strided(dz, 1, E, xc, 1, K, E, K, M)   # dW[j,k] = sum_m dz[m,j] x[m,k]
        dx = ops.f32_gemm_strided(dz, E, 1, W, 1, K, M, K, E)    # dx[m,k] = sum_j dz[m,j] W[j,k]
        db = ops.colsum(dz) if has_bias else None
        return dx, dW, db, None


l2norm_fwd_op = _define("l2_normalize_fwd", "(Tensor x) -> Tensor", lambda x: ops.l2_normalize(x), lambda x: torch.empty_like(x))
l2norm_bwd_op = _define("l2_normalize_bwd", "(Tensor x, Tensor dy) -> Tensor", lambda x, dy: ops.l2_normalize_bwd(x, dy),
                        lambda x, dy: torch.empty_like(x))


class L2NormalizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xc = c32(x)
        ctx.save_for_backward(xc)
        return l2norm_fwd_op(xc)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return l2norm_bwd_op(x, dy.contiguous())


def bias_or_zeros(linear) -> "torch.Tensor":
    """A Linear's bias, or -- for the reference's `add_bias=False` attention projections (modules/layers/multi_head_attention.py:107-113,
    modules/layers/attention.py:100-113) -- a cached fp32 zero vector that requires no grad: the training kernels add it (exactly nothing) and the
    gradient the backward returns for it is dropped by autograd.  A plain attribute, not a buffer: state_dict stays the reference's."""
    if linear.bias is not None:
        return linear.bias
    z = getattr(linear, "_mmamd_zero_bias", None)
    if z is None or z.device != linear.weight.device or z.shape[0] != linear.out_features:
        z = torch.zeros(linear.out_features, dtype=torch.float32, device=linear.weight.device)
        object.__setattr__(linear, "_mmamd_zero_bias", z)
    return z


def plain_layers(layers, cls) -> bool:
    """True when every member of a layer stack is EXACTLY `cls` and carries no hooks: the stack may then read the layers' parameters
    directly (one autograd node, grouped launches, in-place residual streams for the whole stack).  Anything else -- a layer wrapped by
    FullyShardedDataParallel or checkpoint_wrapper (reference examples/flava/native/train.py:141-206 wraps exactly the encoder layers), a
    layer with forward / backward hooks, a subclass with its own forward -- must be CALLED as a module, one layer at a time: FSDP gathers a
    layer's parameters only around that layer's own forward() (outside it they are views of a freed buffer), and hooks fire on __call__."""
    for m in layers:
        if type(m) is not cls:
            return False
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
            return False
    return True


def grad_requested(module, *inputs) -> bool:
    """autograd would record this forward in the reference: grad mode is on and a parameter or an input requires grad."""
    if not torch.is_grad_enabled():
        return False
    if any(isinstance(t, torch.Tensor) and t.requires_grad for t in inputs):
        return True
    return params_require_grad(module)


def params_require_grad(module) -> bool:
    """A parameter of `module` (or of a sub-module) requires grad -- including parameters FullyShardedDataParallel has DE-REGISTERED: with
    use_orig_params=False (FSDP's default, what the reference trainer uses) a wrapped unit's modules hold, inside the unit's forward, plain
    Tensor VIEWS of the flat parameter as ordinary attributes (module.__dict__), and module.parameters() yields nothing.  Missing them sent
    such modules down the inference path: no gradient for their weights (found by tests/test_gpu_fsdp_single_rank.py: FLAVA's cls_token /
    embeddings were never updated)."""
    for p in module.parameters():
        if p.requires_grad:
            return True
    for m in module.modules():  # only reached when no REGISTERED parameter requires grad
        for v in m.__dict__.values():
            if isinstance(v, torch.Tensor) and v.requires_grad:
                return True
    return False


_EVAL_GRAD_MSG = ("{name}: this eval-mode forward has no differentiable path on the MI355X kernels (the differentiable path of the module runs "
                  "in train mode and returns no attention probabilities) and an INPUT requires grad, so the caller expects gradients to flow: "
                  "detach the input / run under torch.no_grad() for inference, or call .train() to differentiate")
_warned_detached = set()


def _inputs_want_grad(*inputs) -> bool:
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in inputs)


def _warn_detached_once(module) -> None:
    """The default state of a freshly built model — eval(), grad mode on, parameters with requires_grad=True — is plain inference in the
    reference's tests and examples: serve it (outputs carry no graph), and say so once per module class instead of refusing the call."""
    name = type(module).__name__
    if name not in _warned_detached:
        _warned_detached.add(name)
        import warnings

        warnings.warn(f"{name}: eval-mode forward with grad mode on returns outputs that are NOT attached to the autograd graph on the MI355X "
                      "path (inference); use torch.no_grad() to silence this, or .train() for the differentiable path", stacklevel=3)


def wants_grad(module, *inputs) -> bool:
    """True: take the differentiable (autograd-node) path.  Modules whose differentiable path changes what they return (FLAVA /
    CoCa: no attention probabilities, only the last hidden state attached) take it in train mode only.  In eval mode the forward is
    inference: it raises only when an input tensor requires grad (the caller is asking for gradients this path cannot give), and warns
    once when merely the parameters do (the default state of any freshly constructed model, in which the reference's tests and examples
    call eval-mode forwards without torch.no_grad())."""
    if not grad_requested(module, *inputs):
        return False
    if module.training:
        return True
    if _inputs_want_grad(*inputs):
        raise NotImplementedError(_EVAL_GRAD_MSG.format(name=type(module).__name__))
    _warn_detached_once(module)
    return False


def forbid_detached_forward(module, *inputs) -> None:
    """Stand-alone layer forwards (one attention module, one MLP called outside its stack) have no differentiable path of their own:
    training runs through the stack-level autograd nodes.  Refuse when an INPUT requires grad (gradients are expected to flow through
    this call); with only parameters requiring grad (every freshly built module) serve the call as inference and warn once."""
    if _inputs_want_grad(*inputs):
        raise NotImplementedError(
            f"{type(module).__name__}: this stand-alone forward has no differentiable path on the MI355X kernels (training goes "
            "through the enclosing encoder / model); call it under torch.no_grad(), or detach its inputs")
    if grad_requested(module, *inputs):
        _warn_detached_once(module)


class CrossEntropyFn(torch.autograd.Function):
    """Mean cross entropy with ignore_index over a small fp32 [N, V] logits matrix (ITM's [B, 2]); gradient in fp32."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index: int):
        lg = logits.detach()
        lg = lg if lg.is_contiguous() else lg.contiguous()
        ctx.save_for_backward(lg, labels)
        ctx.ignore = ignore_index
        return ops.cross_entropy(lg, labels, ignore_index)

    @staticmethod
    def backward(ctx, g):
        lg, labels = ctx.saved_tensors
        return ops.cross_entropy_bwd(lg, labels, ctx.ignore, g.detach().reshape(1).to(f32).contiguous()), None, None


class TanhRowsLinearFn(torch.autograd.Function):
    """Pooler: tanh(x2d[rows] . W^T + b) (modules/losses/flava.py:84-97), differentiable."""

    @staticmethod
    def forward(ctx, x2d, rows64, weight, bias):
        d = x2d.shape[1]
        xc = x2d.detach()
        xc = xc if xc.is_contiguous() else xc.contiguous()
        sel = ops.gather_rows(xc, d, rows64.to(torch.int32), d, f32)
        y = ops.rows_linear_f32(sel, d, sel.shape[0], c32(weight), c32(bias), tanh=True)
        ctx.save_for_backward(sel, rows64, weight, y)
        ctx.xshape = tuple(x2d.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        sel, rows64, weight, y = ctx.saved_tensors
        B, d = sel.shape
        E = y.shape[1]
        # dz = dy * (1 - y^2): a [B, E] elementwise product — host-level glue on the pooled rows (B x 768 values)
        dz = (dy.detach() * (1.0 - y * y)).contiguous()
        W = c32(weight)
        dW = ops.f32_gemm_strided(dz, 1, E, sel, 1, d, E, d, B)
        dsel = ops.f32_gemm_strided(dz, E, 1, W, 1, d, B, d, E)
        db = ops.colsum(dz)
        dx = torch.zeros(ctx.xshape, dtype=f32, device=dz.device)
        ops.scatter_add_rows_(dx, rows64, dsel)
        return dx, None, dW, db


class MaskedHeadLossFn(torch.autograd.Function):
    """MaskedPredictionLoss on the labelled positions (modules/losses/flava.py:174-238): rows -> dense -> GELU -> LayerNorm ->
    vocabulary projection (+tied bias) -> mean cross entropy.  Returns (loss, logits [Nm, V]); the logits are an output, not
    differentiable through this node."""

    @staticmethod
    def forward(ctx, base, idx32, labels, dense_w, dense_b, ln_w, ln_b, dec_w, dec_b, eps: float, ignore_index: int):
        B, S, d = base.shape
        V = dec_w.shape[0]
        b2d = base.detach()
        b2d = (b2d if b2d.is_contiguous() else b2d.contiguous()).view(B * S, d)
        rows = ops.gather_rows(b2d, d, idx32, d, bf)
        u = ops.gemm_bf16(rows, ops.convert(c32(dense_w), bf), c32(dense_b))           # pre-activation, bf16
        g = ops.convert(ops.act_fwd(u, ops.ACT_GELU_ERF), f32)                          # LayerNorm input (fp32 for its backward)
        n = ops.layernorm(g, c32(ln_w), c32(ln_b), eps, out_dtype=bf)
        Vp = (V + 63) // 64 * 64
        wp = torch.zeros((Vp, d), dtype=bf, device=base.device)                          # vocabulary padded to the GEMM granule
        ops.convert(c32(dec_w), bf, out=wp[:V])
        bp = torch.zeros(Vp, dtype=f32, device=base.device)
        bp[:V].copy_(c32(dec_b))
        logits = ops.gemm_bf16(n, wp, bp, out_dtype=f32)[:, :V]
        loss = ops.cross_entropy(logits, labels, ignore_index)
        ctx.save_for_backward(rows, u, g, n, logits, labels, idx32, dense_w, ln_w, wp)
        ctx.meta = (tuple(base.shape), eps, ignore_index, V, Vp)
        ctx.mark_non_differentiable(logits)
        return loss, logits

    @staticmethod
    def backward(ctx, g_loss, _g_logits):
        rows, u, g, n, logits, labels, idx32, dense_w, ln_w, wp = ctx.saved_tensors
        (B, S, d), eps, ignore_index, V, Vp = ctx.meta
        dlog = ops.cross_entropy_bwd(logits, labels, ignore_index, g_loss.detach().reshape(1).to(f32).contiguous(), out_dtype=bf,
                                     pad_cols_to=64)                                    # bf16 [Nm, Vp]
        # logits = n Wdec^T + bias
        dn = ops.gemm_bf16(dlog, ops.transpose_to_bf16(wp, pad_to=64), None, out_dtype=f32)  # [Nm, d] = dlog . Wdec
        dWp, dbp = wgrad(dlog, n, bias=True)                                             # [Vp, d], [Vp]
        dg_, dlnw, dlnb = ops.layernorm_bwd(g, c32(ln_w), dn, eps)
        du = ops.act_bwd(u, ops.convert(dg_, bf), ops.ACT_GELU_ERF)
        drows = dgrad(du, c32(dense_w), f32)
        dWd, dbd = wgrad(du, rows, bias=True)
        dbase = torch.zeros((B * S, d), dtype=f32, device=drows.device)                  # memset; only labelled rows get gradient
        ops.scatter_add_rows_(dbase, idx32.to(torch.int64), drows)
        return dbase.view(B, S, d), None, None, dWd, dbd, dlnw, dlnb, dWp[:V], dbp[:V], None, None


class LinearFn(torch.autograd.Function):
    """y[M,N] (fp32) = x[M,K] W^T (+ b) on the bf16 MFMA GEMM, differentiable (dgrad / split-K wgrad)."""

    @staticmethod
    def forward(ctx, x2d, weight, bias):
        xb = ops.convert(x2d.detach() if x2d.is_contiguous() else x2d.detach().contiguous(), bf)
        N = weight.shape[0]
        Np = (N + 7) // 8 * 8
        wb = ops.convert(c32(weight), bf)
        bb = c32(bias) if bias is not None else None
        if Np != N:  # GEMM granule: N % 8
            wp = torch.zeros((Np, weight.shape[1]), dtype=bf, device=xb.device)
            wp[:N].copy_(wb)
            wb = wp
            if bb is not None:
                bp = torch.zeros(Np, dtype=f32, device=xb.device)
                bp[:N].copy_(bb)
                bb = bp
        y = ops.gemm_bf16(xb, wb, bb, out_dtype=f32)
        ctx.save_for_backward(xb, weight)
        ctx.has_bias = bias is not None
        return y if Np == N else y[:, :N]

    @staticmethod
    def backward(ctx, dy):
        xb, weight = ctx.saved_tensors
        dyc = dy.detach()
        dyb = ops.convert(dyc if dyc.is_contiguous() else dyc.contiguous(), bf)
        dx = dgrad(dyb, c32(weight), f32)
        if ctx.has_bias:
            dW, db = wgrad(dyb, xb, bias=True)
        else:
            dW, db = wgrad(dyb, xb), None
        return dx, dW, db


class CrossAttentionFn(torch.autograd.Function):
    """MultiHeadAttentionWithCache(query, kv, kv) without cache (modules/layers/multi_head_attention.py:115-180) for 2-D token
    matrices: q_in fp32 [B*Sq, dq] — or [Sq, dq] when shared by every sample (AttentionPooler) — and kv_in fp32 [B*Sk, dkv]
    -> fp32 [B*Sq, dq].  One GEMM for q, one for [k | v], the general attention kernel, the output GEMM; backward likewise."""

    @staticmethod
    def forward(ctx, q_in, kv_in, B: int, Sq: int, Sk: int, H: int, shared_q: bool, wq, bq, wk, bk, wv, bv, wo, bo):
        dq = wq.shape[0]
        hd = dq // H
        qb_in = ops.convert(q_in.detach().contiguous(), bf)
        kvb_in = ops.convert(kv_in.detach().contiguous(), bf)
        has_b = bq is not None
        q = ops.gemm_bf16(qb_in, ops.convert(c32(wq), bf), c32(bq) if has_b else None)
        wkv = torch.cat([c32(wk), c32(wv)], 0)
        bkv = torch.cat([c32(bk), c32(bv)], 0) if has_b else None
        kv = ops.gemm_bf16(kvb_in, ops.convert(wkv, bf), bkv)
        lse = torch.empty((B, H, Sq), dtype=f32, device=q.device)
        att, _ = ops.attention_x_fwd(q, kv[:, :dq], kv[:, dq:], B, Sq, Sk, H, hd, None, shared_q=shared_q, lse=lse)
        y = ops.gemm_bf16(att, ops.convert(c32(wo), bf), c32(bo), out_dtype=f32)
        ctx.save_for_backward(qb_in, kvb_in, q, kv, att, lse, wq, wk, wv, wo)
        ctx.meta = (B, Sq, Sk, H, hd, shared_q, has_b)
        return y

    @staticmethod
    def backward(ctx, dy):
        qb_in, kvb_in, q, kv, att, lse, wq, wk, wv, wo = ctx.saved_tensors
        B, Sq, Sk, H, hd, shared_q, has_b = ctx.meta
        dq_ = wq.shape[0]
        dyb = ops.convert(dy.detach().contiguous(), bf)
        datt = dgrad(dyb, c32(wo), bf)
        dWo, dbo = wgrad(dyb, att, bias=True)
        dqa, dkv = ops.attention_x_bwd(q, kv[:, :dq_], kv[:, dq_:], att, datt, lse, B, Sq, Sk, H, hd, None, shared_q=shared_q)
        if shared_q:  # the same queries for every sample: their gradient is the sum over the batch
            dqa = ops.convert(ops.colsum(dqa.view(B, Sq * dq_)).view(Sq, dq_), bf)
        dq_in = dgrad(dqa, c32(wq), f32)
        wkv = torch.cat([c32(wk), c32(wv)], 0)
        dkv_in = dgrad(dkv, wkv, f32)
        if has_b:
            dWq, dbq = wgrad(dqa, qb_in, bias=True)
            dWkv, dbkv = wgrad(dkv, kvb_in, bias=True)
            dbk, dbv = dbkv[:dq_], dbkv[dq_:]
        else:
            dWq, dbq = wgrad(dqa, qb_in), None
            dWkv, dbk, dbv = wgrad(dkv, kvb_in), None, None
        return dq_in, dkv_in, None, None, None, None, None, dWq, dbq, dWkv[:dq_], dbk, dWkv[dq_:], dbv, dWo, dbo


class DecoderStackConfig:
    """Static description for DecoderStackFn.  layers: list of dicts with n_head, eps (attention / cross / feedforward), act,
    has_cross; params per layer in the order self q/k/v/o (w, b), attention LN (w, b), [cross q/k/v/o (w, b), cross LN (w, b)],
    ff0 (w, b), ff1 (w, b), feedforward LN (w, b)."""

    def __init__(self, B: int, S: int, Sk: int, layers, mask: Optional[ops.AttnMask], drop_p: float = 0.0, seed: int = 0, layer0: int = 0,
                 cross_mask: Optional[ops.AttnMask] = None):
        self.B, self.S, self.Sk, self.layers, self.mask = B, S, Sk, layers, mask or ops.AttnMask()
        # mask of the cross-attention blocks ([S, Sk] per sample or shared): a stand-alone TransformerDecoderLayer takes one (reference transformer.py:
        # 366-376); the reference's TransformerDecoder does not hand its own to its layers (:630-636), so a whole stack runs without
        self.cross_mask = cross_mask
        self.layer0 = int(layer0)  # index of layers[0] in the module's stack (a stack run as one node per layer keeps its dropout sites)
        # training-time dropout: ONE rate on the six sites of a layer (self-attention probabilities and branch, cross-attention probabilities and
        # branch, the MLP's hidden dropout, the feed-forward branch), masks from Philox(seed, 16 * layer + site)
        self.drop_p, self.seed = float(drop_p), int(seed)

    def nparams(self, li: int) -> int:
        return 26 if self.layers[li]["has_cross"] else 16


class DecoderStackFn(torch.autograd.Function):
    """TransformerDecoder layers, pre-norm (modules/layers/transformer.py:398-433) or post-norm (:435-470; layer spec "post"): self-attention with the
    given mask, optional cross-attention to `enc` (fp32 [B*Sk, dkv], differentiable), feed-forward.  x0 fp32 [B*S, d]."""

    @staticmethod
    def forward(ctx, x0, enc, cfg: DecoderStackConfig, *params):
        B, S, Sk = cfg.B, cfg.S, cfg.Sk
        x = x0.detach()
        x = x if x.is_contiguous() else x.contiguous()
        encb = ops.convert(enc.detach().contiguous(), bf) if enc is not None else None
        # the bf16 operand copies of every Linear of the stack in one launch per 64 matrices (r06: one convert launch per weight and call before --
        # 147 per CoCa step); the fused in-projections [Wq; Wk; Wv] / [Wk; Wv] are concatenated in fp32 first
        mats, slot, off = [], [], 0
        for li, L in enumerate(cfg.layers):
            pr = [c32(t) for t in params[off:off + cfg.nparams(li)]]
            off += cfg.nparams(li)
            ent = {"qkv": len(mats), "o": len(mats) + 1}
            mats += [torch.cat([pr[0], pr[2], pr[4]], 0), pr[6]]
            if L["has_cross"]:
                ent.update(cq=len(mats), ckv=len(mats) + 1, co=len(mats) + 2)
                mats += [pr[10], torch.cat([pr[12], pr[14]], 0), pr[16]]
            ff = pr[20:] if L["has_cross"] else pr[10:]
            ent.update(w1=len(mats), w2=len(mats) + 1)
            mats += [ff[0], ff[2]]
            slot.append(ent)
        wbf, _ = ops.pack_weights(mats, want_nt=True, want_tr=False)
        recs, off = [], 0
        for li, L in enumerate(cfg.layers):
            pr = [c32(t) for t in params[off:off + cfg.nparams(li)]]
            off += cfg.nparams(li)
            H, d = L["n_head"], x.shape[1]
            hd = d // H
            Wb = {k: wbf[i] for k, i in slot[li].items()}
            qw, qb, kw, kb, vw, vb, ow, ob, g1, be1 = pr[:10]
            pd, sd = cfg.drop_p, cfg.seed  # ONE rate on every dropout site of a decoder layer (reference transformer.py:262-290)
            if L.get("post"):
                # post-norm layer (the reference's DEFAULT, norm_first=False; transformer.py:435-470): every block is LN(x + drop(f(x))) -- the
                # sub-blocks read the fp32 stream itself (rounded to bf16 for the MFMA) and the LayerNorms sit at the END of the blocks.  r05.
                site = 16 * (cfg.layer0 + li)

                def branch(delta_in, w_o, b_o, res, s_):  # res + drop(delta_in W_o^T + b_o)
                    if pd > 0:
                        return ops.dropout(ops.gemm_bf16(delta_in, w_o, b_o, out_dtype=f32), pd, sd, s_, residual=res)
                    return ops.gemm_bf16(delta_in, w_o, b_o, residual=res, out_dtype=f32, out=torch.empty_like(res))

                h1 = ops.convert(x, bf)
                qkv = ops.gemm_bf16(h1, Wb["qkv"], torch.cat([qb, kb, vb], 0))
                lse = torch.empty((B, H, S), dtype=f32, device=x.device)
                att, _ = ops.attention_x_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, S, S, H, hd, cfg.mask, lse=lse, drop=(pd, sd, site + 3))
                a_raw = branch(att, Wb["o"], ob, x, site)
                a = ops.layernorm(a_raw, g1, be1, L["eps1"], out_dtype=f32)
                rec = [x, h1, qkv, att, lse, a_raw, a]
                if L["has_cross"]:
                    cqw, cqb, ckw, ckb, cvw, cvb, cow, cob, gc, bec = pr[10:20]
                    hc = ops.convert(a, bf)
                    qc = ops.gemm_bf16(hc, Wb["cq"], cqb)
                    kvc = ops.gemm_bf16(encb, Wb["ckv"], torch.cat([ckb, cvb], 0))
                    lsec = torch.empty((B, H, S), dtype=f32, device=x.device)
                    attc, _ = ops.attention_x_fwd(qc, kvc[:, :d], kvc[:, d:], B, S, Sk, H, hd, cfg.cross_mask, lse=lsec, drop=(pd, sd, site + 5))
                    c_raw = branch(attc, Wb["co"], cob, a, site + 4)
                    a2 = ops.layernorm(c_raw, gc, bec, L["epsc"], out_dtype=f32)
                    rec += [hc, qc, kvc, attc, lsec, c_raw, a2]
                    ff = pr[20:]
                else:
                    a2 = a
                    ff = pr[10:]
                w1, b1, w2, b2, g2, be2 = ff
                h2 = ops.convert(a2, bf)
                u, g = ops.gemm_bf16_dual(h2, Wb["w1"], b1, L["act"])
                if pd > 0:
                    ops.dropout(g, pd, sd, site + 1, out=g)
                f_raw = branch(g, Wb["w2"], b2, a2, site + 2)
                x = ops.layernorm(f_raw, g2, be2, L["eps2"], out_dtype=f32)
                rec += [h2, u, g, f_raw]
                recs.append(rec)
                continue
            h1 = ops.layernorm(x, g1, be1, L["eps1"], out_dtype=bf)
            qkv = ops.gemm_bf16(h1, Wb["qkv"], torch.cat([qb, kb, vb], 0))
            lse = torch.empty((B, H, S), dtype=f32, device=x.device)
            att, _ = ops.attention_x_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, S, S, H, hd, cfg.mask, lse=lse,
                                         drop=(pd, sd, 16 * (cfg.layer0 + li) + 3))
            if pd > 0:
                a = ops.dropout(ops.gemm_bf16(att, Wb["o"], ob, out_dtype=f32), pd, sd, 16 * (cfg.layer0 + li), residual=x)
            else:
                a = ops.gemm_bf16(att, Wb["o"], ob, residual=x, out_dtype=f32, out=torch.empty_like(x))
            rec = [x, h1, qkv, att, lse, a]
            if L["has_cross"]:
                cqw, cqb, ckw, ckb, cvw, cvb, cow, cob, gc, bec = pr[10:20]
                hc = ops.layernorm(a, gc, bec, L["epsc"], out_dtype=bf)
                qc = ops.gemm_bf16(hc, Wb["cq"], cqb)
                kvc = ops.gemm_bf16(encb, Wb["ckv"], torch.cat([ckb, cvb], 0))
                lsec = torch.empty((B, H, S), dtype=f32, device=x.device)
                attc, _ = ops.attention_x_fwd(qc, kvc[:, :d], kvc[:, d:], B, S, Sk, H, hd, cfg.cross_mask, lse=lsec, drop=(pd, sd, 16 * (cfg.layer0 + li) + 5))
                if pd > 0:
                    a2 = ops.dropout(ops.gemm_bf16(attc, Wb["co"], cob, out_dtype=f32), pd, sd, 16 * (cfg.layer0 + li) + 4, residual=a)
                else:
                    a2 = ops.gemm_bf16(attc, Wb["co"], cob, residual=a, out_dtype=f32, out=torch.empty_like(x))
                rec += [hc, qc, kvc, attc, lsec, a2]
                ff = pr[20:]
            else:
                a2 = a
                ff = pr[10:]
            w1, b1, w2, b2, g2, be2 = ff
            h2 = ops.layernorm(a2, g2, be2, L["eps2"], out_dtype=bf)
            u, g = ops.gemm_bf16_dual(h2, Wb["w1"], b1, L["act"])
            if pd > 0:
                ops.dropout(g, pd, sd, 16 * (cfg.layer0 + li) + 1, out=g)
                x_out = ops.dropout(ops.gemm_bf16(g, Wb["w2"], b2, out_dtype=f32), pd, sd, 16 * (cfg.layer0 + li) + 2, residual=a2)
            else:
                x_out = ops.gemm_bf16(g, Wb["w2"], b2, residual=a2, out_dtype=f32, out=torch.empty_like(x))
            rec += [h2, u, g]
            recs.append(rec)
            x = x_out
        flat = [t for rec in recs for t in rec]
        ctx.save_for_backward(*flat, *params, *([encb] if encb is not None else []))
        ctx.cfg, ctx.nparam, ctx.counts, ctx.has_enc = cfg, len(params), [len(r) for r in recs], encb is not None
        return x

    @staticmethod
    def backward(ctx, dx_out):
        cfg, nparam, counts, has_enc = ctx.cfg, ctx.nparam, ctx.counts, ctx.has_enc
        tensors = list(ctx.saved_tensors)
        encb = tensors.pop() if has_enc else None
        params = tensors[len(tensors) - nparam:]
        flat = tensors[:len(tensors) - nparam]
        recs, o = [], 0
        for c in counts:
            recs.append(flat[o:o + c])
            o += c
        B, S, Sk = cfg.B, cfg.S, cfg.Sk
        dX = dx_out.detach()
        dX = dX if dX.is_contiguous() else dX.contiguous()
        d_enc = None
        grads: List[Optional[Tensor]] = [None] * nparam
        offs, o = [], 0
        for li in range(len(cfg.layers)):
            offs.append(o)
            o += cfg.nparams(li)
        dXb = None
        pending = _pending()  # parked LayerNorm-backward reductions: one launch at the end (ops.colsum_flush)
        # the bf16 TRANSPOSES the dgrad GEMMs read (dX = dY . W as an NT GEMM against W^T), every Linear of the stack in one launch per 64 matrices
        # (r06: one transpose launch per dgrad before -- 171 per CoCa step, 4.4 ms)
        mats, tslot = [], []
        for li, L in enumerate(cfg.layers):
            pr = [c32(t) for t in params[offs[li]:offs[li] + cfg.nparams(li)]]
            ent = {"q": len(mats), "kv": len(mats) + 1, "o": len(mats) + 2}
            mats += [pr[0], torch.cat([pr[2], pr[4]], 0), pr[6]]
            if L["has_cross"]:
                ent.update(cq=len(mats), ckv=len(mats) + 1, co=len(mats) + 2)
                mats += [pr[10], torch.cat([pr[12], pr[14]], 0), pr[16]]
            ff = pr[20:] if L["has_cross"] else pr[10:]
            ent.update(w1=len(mats), w2=len(mats) + 1)
            mats += [ff[0], ff[2]]
            tslot.append(ent)
        _, wtr = ops.pack_weights(mats, want_nt=False, want_tr=True)
        for li in reversed(range(len(cfg.layers))):
            L, rec = cfg.layers[li], recs[li]
            pr = [c32(t) for t in params[offs[li]:offs[li] + cfg.nparams(li)]]
            H = L["n_head"]
            WT = {k: wtr[i] for k, i in tslot[li].items()}
            J = {}  # the layer's weight-gradient jobs (dY, X), run as ONE grouped split-K launch + one reduce at the end of the layer (r06; eight launch pairs before)
            qw, qb, kw, kb, vw, vb, ow, ob, g1, be1 = pr[:10]
            d = qw.shape[0]
            hd = d // H
            if L.get("post"):
                # backward of the post-norm layer: each LayerNorm backward comes FIRST in its block, and the residual sums are epilogues of the dgrad GEMMs
                pd, sd, site = cfg.drop_p, cfg.seed, 16 * (cfg.layer0 + li)
                gl = [None] * cfg.nparams(li)
                h2, u, g, f_raw = rec[-4:]
                w1, b1, w2, b2, g2, be2 = pr[20:] if L["has_cross"] else pr[10:]
                d_f, dg2, dbe2, d_fb = ops.layernorm_bwd(f_raw, g2, dX, L["eps2"], want_bf16=True, defer=pending)
                if pd > 0:
                    d_fb = ops.dropout(d_f, pd, sd, site + 2, out_dtype=bf)
                du = dgrad_t(d_fb, WT["w2"], bf, _ACT_GRAD[L["act"]], u)
                if pd > 0:
                    ops.dropout(du, pd, sd, site + 1, out=du)
                J["w2"] = (d_fb, g)
                d_a2 = dgrad_t(du, WT["w1"], f32, ops.ACT_NONE, d_f)  # d_f + du W1: a2 feeds the MLP and the residual
                J["w1"] = (du, h2)
                if L["has_cross"]:
                    cqw, cqb, ckw, ckb, cvw, cvb, cow, cob, gc, bec = pr[10:20]
                    hc, qc, kvc, attc, lsec, c_raw, _a2 = rec[7:14]
                    d_c, dgc, dbec, d_cb = ops.layernorm_bwd(c_raw, gc, d_a2, L["epsc"], want_bf16=True, defer=pending)
                    if pd > 0:
                        d_cb = ops.dropout(d_c, pd, sd, site + 4, out_dtype=bf)
                    dattc = dgrad_t(d_cb, WT["co"], bf)
                    J["co"] = (d_cb, attc)
                    dqc, dkvc = ops.attention_x_bwd(qc, kvc[:, :d], kvc[:, d:], attc, dattc, lsec, B, S, Sk, H, hd, cfg.cross_mask, drop=(pd, sd, site + 5))
                    d_a = dgrad_t(dqc, WT["cq"], f32, ops.ACT_NONE, d_c)  # d_c + dqc Wq
                    J["cq"] = (dqc, hc)
                    d_enc = dgrad_t(dkvc, WT["ckv"], f32) if d_enc is None else ops.gemm_bf16(dkvc, WT["ckv"], None, residual=d_enc, out_dtype=f32, out=d_enc)
                    J["ckv"] = (dkvc, encb)
                else:
                    d_a = d_a2
                x, h1, qkv, att, lse, a_raw, _a = rec[:7]
                d_ar, dg1, dbe1, d_arb = ops.layernorm_bwd(a_raw, g1, d_a, L["eps1"], want_bf16=True, defer=pending)
                if pd > 0:
                    d_arb = ops.dropout(d_ar, pd, sd, site, out_dtype=bf)
                datt = dgrad_t(d_arb, WT["o"], bf)
                J["o"] = (d_arb, att)
                dq, dkv = ops.attention_x_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], att, datt, lse, B, S, S, H, hd, cfg.mask,
                                              drop=(pd, sd, site + 3))
                dh1 = dgrad_t(dq, WT["q"], f32, ops.ACT_NONE, d_ar)  # d_ar + dq Wq + [dk | dv] [Wk; Wv]: x feeds the attention and the residual
                dh1 = ops.gemm_bf16(dkv, WT["kv"], None, residual=dh1, out_dtype=f32, out=dh1)
                J["q"] = (dq, h1)
                J["kv"] = (dkv, h1)
                G = dict(zip(J.keys(), wgrad_many([(dy_, x_, True) for dy_, x_ in J.values()])))
                (dWq, dbq), (dWkv, dbkv), (dWo, dbo), (dW1, db1), (dW2, db2) = G["q"], G["kv"], G["o"], G["w1"], G["w2"]
                gl[:10] = [dWq, dbq, dWkv[:d], dbkv[:d], dWkv[d:], dbkv[d:], dWo, dbo, dg1, dbe1]
                if L["has_cross"]:
                    (dWcq, dbcq), (dWckv, dbckv), (dWco, dbco) = G["cq"], G["ckv"], G["co"]
                    gl[10:20] = [dWcq, dbcq, dWckv[:d], dbckv[:d], dWckv[d:], dbckv[d:], dWco, dbco, dgc, dbec]
                    gl[20:] = [dW1, db1, dW2, db2, dg2, dbe2]
                else:
                    gl[10:] = [dW1, db1, dW2, db2, dg2, dbe2]
                grads[offs[li]:offs[li] + cfg.nparams(li)] = gl
                dX, dXb = dh1, None
                continue
            x, h1, qkv, att, lse, a = rec[:6]
            if L["has_cross"]:
                hc, qc, kvc, attc, lsec, a2 = rec[6:12]
                h2, u, g = rec[12:15]
                cqw, cqb, ckw, ckb, cvw, cvb, cow, cob, gc, bec = pr[10:20]
                w1, b1, w2, b2, g2, be2 = pr[20:]
            else:
                a2 = a
                h2, u, g = rec[6:9]
                w1, b1, w2, b2, g2, be2 = pr[10:]
            pd, sd = cfg.drop_p, cfg.seed
            if pd > 0:  # the feed-forward branch's gradient is the masked, scaled dX; the residual path keeps dX
                dXb = ops.dropout(dX, pd, sd, 16 * (cfg.layer0 + li) + 2, out_dtype=bf)
            elif dXb is None:
                dXb = ops.convert(dX, bf)
            du = dgrad_t(dXb, WT["w2"], bf, _ACT_GRAD[L["act"]], u)
            if pd > 0:
                ops.dropout(du, pd, sd, 16 * (cfg.layer0 + li) + 1, out=du)
            J["w2"] = (dXb, g)
            dh2 = dgrad_t(du, WT["w1"], f32)
            J["w1"] = (du, h2)
            d_a2, dg2, dbe2, d_a2b = ops.layernorm_bwd(a2, g2, dh2, L["eps2"], add=dX, want_bf16=True, defer=pending)
            gl = [None] * cfg.nparams(li)
            if L["has_cross"]:
                if pd > 0:
                    d_a2b = ops.dropout(d_a2, pd, sd, 16 * (cfg.layer0 + li) + 4, out_dtype=bf)
                dattc = dgrad_t(d_a2b, WT["co"], bf)
                J["co"] = (d_a2b, attc)
                dqc, dkvc = ops.attention_x_bwd(qc, kvc[:, :d], kvc[:, d:], attc, dattc, lsec, B, S, Sk, H, hd, cfg.cross_mask, drop=(pd, sd, 16 * (cfg.layer0 + li) + 5))
                dhc = dgrad_t(dqc, WT["cq"], f32)
                J["cq"] = (dqc, hc)
                d_enc = dgrad_t(dkvc, WT["ckv"], f32) if d_enc is None else ops.gemm_bf16(dkvc, WT["ckv"], None, residual=d_enc, out_dtype=f32, out=d_enc)
                J["ckv"] = (dkvc, encb)
                d_a, dgc, dbec, d_ab = ops.layernorm_bwd(a, gc, dhc, L["epsc"], add=d_a2, want_bf16=True, defer=pending)
            else:
                d_a, d_ab = d_a2, d_a2b
            if pd > 0:
                d_ab = ops.dropout(d_a, pd, sd, 16 * (cfg.layer0 + li), out_dtype=bf)
            datt = dgrad_t(d_ab, WT["o"], bf)
            J["o"] = (d_ab, att)
            dq, dkv = ops.attention_x_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], att, datt, lse, B, S, S, H, hd, cfg.mask,
                                          drop=(pd, sd, 16 * (cfg.layer0 + li) + 3))
            # dh1 = dq Wq + [dk | dv] [Wk; Wv]: two GEMMs, the second accumulates onto the first
            dh1 = dgrad_t(dq, WT["q"], f32)
            dh1 = ops.gemm_bf16(dkv, WT["kv"], None, residual=dh1, out_dtype=f32, out=dh1)
            J["q"] = (dq, h1)
            J["kv"] = (dkv, h1)
            dX, dg1, dbe1, dXb = ops.layernorm_bwd(x, g1, dh1, L["eps1"], add=d_a, want_bf16=True, defer=pending)
            G = dict(zip(J.keys(), wgrad_many([(dy_, x_, True) for dy_, x_ in J.values()])))
            (dWq, dbq), (dWkv, dbkv), (dWo, dbo), (dW1, db1), (dW2, db2) = G["q"], G["kv"], G["o"], G["w1"], G["w2"]
            gl[:10] = [dWq, dbq, dWkv[:d], dbkv[:d], dWkv[d:], dbkv[d:], dWo, dbo, dg1, dbe1]
            if L["has_cross"]:
                (dWcq, dbcq), (dWckv, dbckv), (dWco, dbco) = G["cq"], G["ckv"], G["co"]
                gl[10:20] = [dWcq, dbcq, dWckv[:d], dbckv[:d], dWckv[d:], dbckv[d:], dWco, dbco, dgc, dbec]
                gl[20:] = [dW1, db1, dW2, db2, dg2, dbe2]
            else:
                gl[10:] = [dW1, db1, dW2, db2, dg2, dbe2]
            grads[offs[li]:offs[li] + cfg.nparams(li)] = gl
        ops.colsum_flush(pending)
        return (dX, d_enc, None, *grads)
