"""Gradients through a STAND-ALONE TransformerDecoderLayer with a cross-attention mask (modules/layers/transformer.py:366-376,395-470: the layer hands
`cross_attention_mask` to its cross-attention's scaled_dot_product_attention; the reference's TransformerDecoder does not forward its own, :630-636) from the
REFERENCE:   python -m tests.golden.make_golden_decoder_xmask_grad  ->  decoder_xmask_grad.npz
  pre.* / post.*   TransformerDecoderLayer(d_model=128, n_head=2, dim_feedforward=256, GELU, eps 1e-5, use_cross_attention=True, dim_kv=64,
                   norm_first=True / False) in TRAIN mode (dropout 0): x [2, 9, 128], encoder states [2, 5, 64], causal self-attention mask, boolean
                   cross-attention mask [2, 1, 9, 5] (per sample; every query keeps at least one key); loss = sum(out * w) -> the gradients of x, of
                   the encoder states and of every parameter"""
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
    from torchmultimodal.modules.layers.transformer import TransformerDecoderLayer

    torch.set_num_threads(8)
    st = {}
    g = torch.Generator().manual_seed(47)
    causal = torch.ones(9, 9, dtype=torch.bool).tril()
    xmask = torch.rand(2, 1, 9, 5, generator=g) > 0.4
    xmask[..., 0] |= ~xmask.any(-1)  # no query without a key
    st["xmask"] = xmask.numpy().copy()
    for tag, nf in (("pre", True), ("post", False)):
        seed(93 if nf else 94)
        layer = TransformerDecoderLayer(d_model=128, n_head=2, dim_feedforward=256, activation=nn.GELU, layer_norm_eps=1e-5, norm_first=nf,
                                        use_cross_attention=True, dim_kv=64).train()
        x = torch.randn(2, 9, 128, generator=g).requires_grad_(True)
        enc = torch.randn(2, 5, 64, generator=g).requires_grad_(True)
        w = torch.randn(2, 9, 128, generator=g)
        y, _ = layer(x, enc, attention_mask=causal, cross_attention_mask=xmask)
        (y * w).sum().backward()
        # the mask is live: without it the output differs
        y0, _ = layer(x, enc, attention_mask=causal)
        assert (y0 - y).abs().max() > 1e-3
        st.update({f"{tag}.x": tnp(x), f"{tag}.w": tnp(w), f"{tag}.y": tnp(y), f"{tag}.dx": tnp(x.grad), f"{tag}.enc": tnp(enc), f"{tag}.denc": tnp(enc.grad)})
        st.update({f"{tag}.sd." + k: v for k, v in sd_np(layer).items()})
        st.update({f"{tag}.g." + k: tnp(p.grad) for k, p in layer.named_parameters()})
    np.savez_compressed(OUT / "decoder_xmask_grad.npz", **st)
    print("decoder_xmask_grad.npz", {k: v.shape for k, v in st.items() if ".sd." not in k and ".g." not in k})


if __name__ == "__main__":
    main()
