"""Gradients through POST-NORM decoder layers (the reference's default, norm_first=False; modules/layers/transformer.py:289,435-470) from the REFERENCE:
python -m tests.golden.make_golden_decoder_post_grad  ->  decoder_post_grad.npz
  dec.*    TransformerDecoder(n_layer=2, d_model=128, n_head=2, dim_feedforward=256, GELU, eps 1e-5, norm_first=False, use_cross_attention=True,
           dim_kv=64) in TRAIN mode (dropout 0): x [2, 9, 128], encoder states [2, 5, 64], causal mask; loss = sum(last_hidden_state * w) ->
           the gradients of x, of the encoder states and of every parameter
  self.*   the same without cross-attention (use_cross_attention=False), one layer, plus final_layer_norm"""
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
    from torchmultimodal.modules.layers.transformer import TransformerDecoder

    torch.set_num_threads(8)
    st = {}
    g = torch.Generator().manual_seed(31)
    causal = torch.ones(9, 9, dtype=torch.bool).tril()
    for tag, kw, with_enc in (("dec", dict(n_layer=2, use_cross_attention=True, dim_kv=64), True),
                              ("self", dict(n_layer=1, use_cross_attention=False, final_layer_norm_eps=1e-5), False)):
        seed(91 if with_enc else 92)
        dec = TransformerDecoder(d_model=128, n_head=2, dim_feedforward=256, activation=nn.GELU, layer_norm_eps=1e-5, norm_first=False, **kw).train()
        x = torch.randn(2, 9, 128, generator=g).requires_grad_(True)
        enc = torch.randn(2, 5, 64, generator=g).requires_grad_(True) if with_enc else None
        w = torch.randn(2, 9, 128, generator=g)
        y = dec(x, enc, attention_mask=causal).last_hidden_state
        (y * w).sum().backward()
        st.update({f"{tag}.x": tnp(x), f"{tag}.w": tnp(w), f"{tag}.y": tnp(y), f"{tag}.dx": tnp(x.grad)})
        if with_enc:
            st.update({f"{tag}.enc": tnp(enc), f"{tag}.denc": tnp(enc.grad)})
        st.update({f"{tag}.sd." + k: v for k, v in sd_np(dec).items()})
        st.update({f"{tag}.g." + k: tnp(p.grad) for k, p in dec.named_parameters()})
    np.savez_compressed(OUT / "decoder_post_grad.npz", **st)
    print("decoder_post_grad.npz", {k: v.shape for k, v in st.items() if ".sd." not in k and ".g." not in k})


if __name__ == "__main__":
    main()
