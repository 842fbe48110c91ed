"""TEST INFRASTRUCTURE (oracle): the reference's pre-norm encoder layer with TRAINING-TIME dropout, restated on torch CPU autograd with the masks
GIVEN (so that a run on the MI355X kernels and this restatement drop the same elements).  Only tests/ may import this.

Arithmetic restated (reference file:line):
  torchmultimodal/modules/layers/transformer.py:80-116   x1 = x + drop_a(attn(LN1(x)));  x2 = x1 + drop_f(mlp(LN2(x1)))
  torchmultimodal/modules/layers/transformer.py:64-70    drop_* = nn.Dropout(p) or ONE StochasticDepth(p, "row") on both branches
  torchmultimodal/modules/layers/mlp.py:52-61            mlp = Linear -> activation -> Dropout(p) -> Linear
  torchmultimodal/modules/layers/multi_head_attention.py:54-77   packed input_proj, SDPA without dropout, output_proj
Masks: oracle/philox.py with the site numbering of multimodal_amd/_autograd.py (16 * layer + {0: attention branch, 1: MLP hidden, 2: feed-forward
branch}); stochastic depth uses one decision per sample (group = S * d elements).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import philox


def _mask(shape, p, seed, site, group=0):
    n = int(np.prod(shape))
    return torch.from_numpy(philox.dropout_mask(n, p, seed, site, group).reshape(shape).astype(np.float32))


def _drop(x, p, seed, site, group=0):
    if p <= 0:
        return x
    scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    return x * _mask(tuple(x.shape), p, seed, site, group) * scale


def encoder_forward(x, layers, n_head, p_branch, p_mlp, path_rates, seed, act="gelu"):
    """x fp32 [B, S, d]; layers: list of dicts with Wqkv [3d,d], bqkv, Wo, bo, W1, b1, W2, b2, g1, be1, g2, be2, eps1, eps2 (torch tensors that
    require grad); path_rates: per-layer stochastic-depth rates or None.  Returns x_L."""
    B, S, d = x.shape
    hd = d // n_head
    for li, L in enumerate(layers):
        rate = path_rates[li] if path_rates is not None else None
        pb, grp = (rate, S * d) if rate is not None else (p_branch, 0)
        h = F.layer_norm(x, (d,), L["g1"], L["be1"], L["eps1"])
        qkv = h @ L["Wqkv"].t() + L["bqkv"]
        q, k, v = (t.view(B, S, n_head, hd).transpose(1, 2) for t in qkv.split(d, dim=-1))
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, S, d)
        x = x + _drop(a @ L["Wo"].t() + L["bo"], pb, seed, 16 * li, grp)
        h = F.layer_norm(x, (d,), L["g2"], L["be2"], L["eps2"])
        u = h @ L["W1"].t() + L["b1"]
        g = F.gelu(u) if act == "gelu" else u * torch.sigmoid(1.702 * u)
        g = _drop(g, p_mlp, seed, 16 * li + 1)
        x = x + _drop(g @ L["W2"].t() + L["b2"], pb, seed, 16 * li + 2, grp)
    return x


def attention_mask_bhqk(B, H, Sq, Sk, p, seed, site):
    """keep mask [B, H, Sq, Sk] of the attention-probability dropout (csrc/common.h::attn_drop_block): element (b, h, q, key) = output word
    key & 3 of the Philox block with 64-bit counter ((b H + h) Sq + q) * ceil(Sk / 4) + key // 4 and (site, 0)."""
    sk4 = (Sk + 3) // 4
    rows = B * H * Sq
    idx = np.arange(rows * sk4, dtype=np.uint64)
    lo, hi = (idx & philox.U32).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    words = np.stack(philox.philox4x32_10(lo, hi, np.full_like(lo, site), np.zeros_like(lo), k0, k1), axis=1).reshape(rows, sk4 * 4)[:, :Sk]
    return (words >= np.uint32(philox.threshold(p))).reshape(B, H, Sq, Sk)


def flava_encoder_forward(x, layers, n_head, p, seed, key_mask=None):
    """FLAVA's pre-norm encoder layer with ONE dropout rate p on all four sites (reference models/flava/transformer.py:87-90: SelfAttention(p),
    attention_dropout, the MLP's hidden dropout, feedforward_dropout): dropout on the softmax output before the product with V
    (modules/layers/attention.py:234-239), masks from the Philox restatement with the kernels' site numbering (16 * layer + {0, 1, 2, 3})."""
    B, S, d = x.shape
    hd = d // n_head
    scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    for li, L in enumerate(layers):
        h = F.layer_norm(x, (d,), L["g1"], L["be1"], L["eps1"])
        qkv = h @ L["Wqkv"].t() + L["bqkv"]
        q, k, v = (t.view(B, S, n_head, hd).transpose(1, 2) for t in qkv.split(d, dim=-1))
        s = (q @ k.transpose(-1, -2)) / (hd ** 0.5)
        if key_mask is not None:
            s = s.masked_fill(key_mask[:, None, None, :] == 0, float("-inf"))
        pr = torch.softmax(s, dim=-1)
        keep = torch.from_numpy(attention_mask_bhqk(B, n_head, S, S, p, seed, 16 * li + 3).astype(np.float32))
        a = ((pr * keep * scale) @ v).transpose(1, 2).reshape(B, S, d)
        x = x + _drop(a @ L["Wo"].t() + L["bo"], p, seed, 16 * li)
        h = F.layer_norm(x, (d,), L["g2"], L["be2"], L["eps2"])
        g = _drop(F.gelu(h @ L["W1"].t() + L["b1"]), p, seed, 16 * li + 1)
        x = x + _drop(g @ L["W2"].t() + L["b2"], p, seed, 16 * li + 2)
    return x
