"""Gradient fixtures from the REFERENCE for the training-path holes VERDICT r04 lists (missing 4 / next 6):
python -m tests.golden.make_golden_layer_grad  ->  layer_grad.npz
  post.*   modules/layers/transformer.py TransformerEncoder, 2 POST-norm layers (norm_first=False, the reference's default, :56,118-132),
           128 wide, 2 heads, GELU, final LayerNorm: forward, hidden states, every parameter gradient and the input gradient
  fpost.*  models/flava/transformer.py TransformerEncoder, 1 post-norm layer with a key-padding attention mask
  mask.*   modules/layers/transformer.py TransformerEncoder, 2 pre-norm layers, trained under a boolean [S, S] attention mask (causal) and
           under a random [B, S, S] mask (:191 raised until r05)
  lone.*   ONE modules/layers/transformer.py TransformerEncoderLayer (pre-norm) and ONE flava TransformerEncoderLayer (post-norm) called stand-alone in
           training: output, parameter gradients, input gradient
Loss in every case: sum(y * w) with a fixed random w (a non-trivial gradient through the final LayerNorm)."""
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


def run(st, tag, mod, x, w, call, store_sd=True):
    mod.train()
    xg = x.clone().requires_grad_(True)
    y = call(mod, xg)
    (y * w).sum().backward()
    st.update({f"{tag}.x": tnp(x), f"{tag}.w": tnp(w), f"{tag}.y": tnp(y), f"{tag}.dx": tnp(xg.grad)})
    if store_sd:
        st.update({f"{tag}.sd." + k: v for k, v in sd_np(mod).items()})
    st.update({f"{tag}.g." + k: tnp(p.grad) for k, p in mod.named_parameters()})
    mod.zero_grad()


def main():
    _ref_shim.install()
    from torch import nn
    from torchmultimodal.models.flava.transformer import TransformerEncoder as FlavaEncoder, TransformerEncoderLayer as FlavaLayer
    from torchmultimodal.modules.layers.transformer import TransformerEncoder, TransformerEncoderLayer

    torch.set_num_threads(8)
    st = {}
    g = torch.Generator().manual_seed(23)
    seed(71)
    enc = TransformerEncoder(n_layer=2, d_model=128, n_head=2, dim_feedforward=128, activation=nn.GELU, layer_norm_eps=1e-5, norm_first=False,
                             final_layer_norm_eps=1e-5)
    x, w = torch.randn(3, 12, 128, generator=g), torch.randn(3, 12, 128, generator=g)
    run(st, "post", enc, x, w, lambda m, t: m(t).last_hidden_state)
    enc.train()
    hs = enc(x, return_hidden_states=True).hidden_states
    st["post.hidden"] = np.stack([tnp(t) for t in hs])

    seed(72)
    fenc = FlavaEncoder(n_layer=1, d_model=128, n_head=2, dim_feedforward=128, activation=nn.GELU, layer_norm_eps=1e-5, norm_first=False)
    am = (torch.rand(3, 1, 1, 12, generator=g) > 0.25).long()
    am[..., 0] = 1
    st["fpost.mask"] = tnp(am)
    x, w = torch.randn(3, 12, 128, generator=g), torch.randn(3, 12, 128, generator=g)
    # the reference's additive convention: flava passes (1 - mask) * -10000-style masks upstream; SelfAttention here takes "0 = do not attend"
    run(st, "fpost", fenc, x, w, lambda m, t: m(t, attention_mask=am).last_hidden_state)

    seed(73)
    menc = TransformerEncoder(n_layer=2, d_model=128, n_head=2, dim_feedforward=128, activation=nn.GELU, layer_norm_eps=1e-5, norm_first=True)
    x, w = torch.randn(2, 10, 128, generator=g), torch.randn(2, 10, 128, generator=g)
    causal = torch.ones(10, 10, dtype=torch.bool).tril()
    run(st, "mask.causal", menc, x, w, lambda m, t: m(t, attention_mask=causal).last_hidden_state)
    rnd = torch.rand(2, 10, 10, generator=g) > 0.35
    rnd[:, :, 0] = True
    st["mask.rnd.mask"] = tnp(rnd)
    run(st, "mask.rnd", menc, x, w, lambda m, t: m(t, attention_mask=rnd.unsqueeze(1)).last_hidden_state, store_sd=False)  # (weights: mask.causal.sd.*)

    seed(74)
    lay = TransformerEncoderLayer(d_model=128, n_head=2, dim_feedforward=128, activation=nn.GELU, layer_norm_eps=1e-5, norm_first=True)
    x, w = torch.randn(2, 9, 128, generator=g), torch.randn(2, 9, 128, generator=g)
    run(st, "lone.pre", lay, x, w, lambda m, t: m(t))
    seed(75)
    flay = FlavaLayer(d_model=128, n_head=2, dim_feedforward=128, activation=nn.GELU, layer_norm_eps=1e-5, norm_first=False)
    run(st, "lone.fpost", flay, x, w, lambda m, t: m(t))
    np.savez_compressed(OUT / "layer_grad.npz", **st)
    print("layer_grad.npz", {k: v.shape for k, v in st.items() if ".sd." not in k and ".g." not in k})


if __name__ == "__main__":
    main()
