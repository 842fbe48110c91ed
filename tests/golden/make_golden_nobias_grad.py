"""Gradient fixture from the REFERENCE for attention projections WITHOUT bias in training (VERDICT r05 missing 4 / next 7):
python -m tests.golden.make_golden_nobias_grad  ->  nobias_grad.npz
  dec.*   ONE modules/layers/transformer.py TransformerDecoderLayer (post-norm, causal mask, cross-attention over a 64-wide memory) whose self- and
          cross-attention are MultiHeadAttentionWithCache(add_bias=False) (modules/layers/multi_head_attention.py:107-113: q / k / v projections
          without bias; the output projection keeps its own).  (TransformerEncoderLayer's MultiHeadSelfAttention has no such option.)
  flava.* ONE models/flava/transformer.py TransformerEncoderLayer whose attention is a modules/layers/attention.py MultiHeadAttention(add_bias=False)
Output, input gradient(s) and every parameter gradient; loss sum(y * w) with a fixed random w."""
from __future__ import annotations

import sys
import warnings
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests.golden import _ref_shim  # noqa: E402
from tests.golden.make_golden import sd_np, seed  # noqa: E402

OUT = Path(__file__).resolve().parent
warnings.filterwarnings("ignore")


def tnp(t):
    return t.detach().numpy().copy()


def main():
    _ref_shim.install()
    from torch import nn
    from torchmultimodal.models.flava.transformer import TransformerEncoderLayer as FlavaLayer
    from torchmultimodal.modules.layers.attention import MultiHeadAttention, SelfAttention
    from torchmultimodal.modules.layers.multi_head_attention import MultiHeadAttentionWithCache
    from torchmultimodal.modules.layers.transformer import TransformerDecoderLayer

    torch.set_num_threads(8)
    st = {}
    g = torch.Generator().manual_seed(31)

    def finish(tag, mod, y, w, inputs):
        (y * w).sum().backward()
        st.update({f"{tag}.w": tnp(w), f"{tag}.y": tnp(y)})
        for name, t in inputs.items():
            st[f"{tag}.{name}"] = tnp(t)
            st[f"{tag}.d{name}"] = tnp(t.grad)
        st.update({f"{tag}.sd." + k: v for k, v in sd_np(mod).items()})
        st.update({f"{tag}.g." + k: tnp(p.grad) for k, p in mod.named_parameters()})

    seed(82)
    dec = TransformerDecoderLayer(d_model=128, n_head=2, dim_feedforward=256, activation=nn.GELU, layer_norm_eps=1e-5, norm_first=False,
                                  use_cross_attention=True, dim_kv=64)
    dec.attention = MultiHeadAttentionWithCache(dim_q=128, dim_kv=128, num_heads=2, add_bias=False)
    dec.cross_attention = MultiHeadAttentionWithCache(dim_q=128, dim_kv=64, num_heads=2, add_bias=False)
    dec.train()
    x = torch.randn(2, 9, 128, generator=g).requires_grad_(True)
    enc = torch.randn(2, 5, 64, generator=g).requires_grad_(True)
    w = torch.randn(2, 9, 128, generator=g)
    causal = torch.ones(9, 9, dtype=torch.bool).tril()
    y, _ = dec(x, enc, attention_mask=causal)
    finish("dec", dec, y, w, {"x": x, "enc": enc})

    seed(83)
    fl = FlavaLayer(d_model=128, n_head=2, dim_feedforward=256, activation=nn.GELU, layer_norm_eps=1e-5, norm_first=True)
    fl.attention = MultiHeadAttention(dim_q=128, dim_kv=128, n_head=2, attn_module=SelfAttention(0.0), add_bias=False)
    assert fl.attention.query.bias is None
    fl.train()
    x = torch.randn(2, 9, 128, generator=g).requires_grad_(True)
    w = torch.randn(2, 9, 128, generator=g)
    finish("flava", fl, fl(x), w, {"x": x})
    np.savez_compressed(OUT / "nobias_grad.npz", **st)
    print("nobias_grad.npz", len(st), "arrays;", {k: v.shape for k, v in st.items() if ".sd." not in k and ".g." not in k})


if __name__ == "__main__":
    main()
