"""Fixtures from the REFERENCE for `head_mask` (modules/layers/attention.py:190,236-237: multiplied into the probabilities after softmax and dropout):
python -m tests.golden.make_golden_head_mask  ->  head_mask.npz
  mha.*   MultiHeadAttention, dim 128, 2 heads, q [3,7,128]: a full [3,2,7,7] 0/1 mask, a per-head [1,2,1,1] mask (head pruning) and a real-valued
          [3,1,7,7] one, each with a key-padding attention_mask; output + returned probabilities
  enc.*   flava TransformerEncoder (2 pre-norm layers, 128 wide, 2 heads) with a per-head mask [1,2,1,1] (reference transformer.py:268-275: the same
          head_mask for every layer): last hidden state, hidden states, attentions
  head_mask_grad.npz (r05): the same encoder in TRAIN mode (all dropout rates 0) with a real-valued [2,2,9,9] head_mask and the key-padding mask:
          loss = sum(last_hidden_state * w) -> the gradient of the input and of every parameter (torch autograd of the reference)
"""
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
    from torchmultimodal.models.flava.transformer import TransformerEncoder
    from torchmultimodal.modules.layers.attention import MultiHeadAttention, SelfAttention

    torch.set_num_threads(8)
    st = {}
    g = torch.Generator().manual_seed(17)
    with torch.no_grad():
        seed(61)
        mha = MultiHeadAttention(dim_q=128, dim_kv=128, n_head=2, attn_module=SelfAttention()).eval()
        x = torch.randn(3, 7, 128, generator=g)
        km = (torch.rand(3, 1, 1, 7, generator=g) > 0.25).long()
        km[..., 0] = 1
        masks = {"full": (torch.rand(3, 2, 7, 7, generator=g) > 0.3).float(), "head": torch.tensor([1.0, 0.0]).view(1, 2, 1, 1),
                 "real": torch.rand(3, 1, 7, 7, generator=g)}
        st.update({"mha.x": tnp(x), "mha.mask": tnp(km)})
        st.update({"mha.sd." + k: v for k, v in sd_np(mha).items()})
        for name, hm in masks.items():
            y, p = mha(x, return_attn_weights=True, attention_mask=km, head_mask=hm)
            st.update({f"mha.{name}.hm": tnp(hm), f"mha.{name}.out": tnp(y), f"mha.{name}.probs": tnp(p)})

        seed(62)
        enc = TransformerEncoder(n_layer=2, d_model=128, n_head=2, dim_feedforward=256, activation=nn.GELU, norm_first=True).eval()
        h = torch.randn(2, 9, 128, generator=g)
        am = (torch.rand(2, 1, 1, 9, generator=g) > 0.2).long()
        am[..., 0] = 1
        hm = torch.tensor([0.0, 1.0]).view(1, 2, 1, 1)
        o = enc(h, attention_mask=am, head_mask=hm, return_attn_weights=True, return_hidden_states=True)
        st.update({"enc.x": tnp(h), "enc.mask": tnp(am), "enc.hm": tnp(hm), "enc.last": tnp(o.last_hidden_state),
                   "enc.hidden": np.stack([tnp(t) for t in o.hidden_states]), "enc.attn": np.stack([tnp(t) for t in o.attentions])})
        st.update({"enc.sd." + k: v for k, v in sd_np(enc).items()})
    np.savez_compressed(OUT / "head_mask.npz", **st)
    print("head_mask.npz", {k: v.shape for k, v in st.items() if ".sd." not in k})

    # ---- gradients through a head-masked encoder (training mode)
    seed(63)
    enc = TransformerEncoder(n_layer=2, d_model=128, n_head=2, dim_feedforward=256, activation=nn.GELU, norm_first=True).train()
    g = torch.Generator().manual_seed(23)
    h = torch.randn(2, 9, 128, generator=g).requires_grad_(True)
    am = (torch.rand(2, 1, 1, 9, generator=g) > 0.2).long()
    am[..., 0] = 1
    hm = torch.rand(2, 2, 9, 9, generator=g)
    hm[0, 1] = 0.0  # one head of one sample pruned outright
    w = torch.randn(2, 9, 128, generator=g)
    o = enc(h, attention_mask=am, head_mask=hm, return_attn_weights=True, return_hidden_states=True)
    loss = (o.last_hidden_state * w).sum()
    loss.backward()
    sg = {"x": tnp(h), "mask": tnp(am), "hm": tnp(hm), "w": tnp(w), "last": tnp(o.last_hidden_state), "loss": tnp(loss), "dx": tnp(h.grad),
          "attn": np.stack([tnp(t) for t in o.attentions])}
    sg.update({"sd." + k: v for k, v in sd_np(enc).items()})
    sg.update({"g." + k: tnp(p.grad) for k, p in enc.named_parameters()})
    np.savez_compressed(OUT / "head_mask_grad.npz", **sg)
    print("head_mask_grad.npz", float(loss), {k: v.shape for k, v in sg.items() if k.startswith("g.")}.__len__(), "parameter gradients")


if __name__ == "__main__":
    main()
