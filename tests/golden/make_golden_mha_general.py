"""Fixtures from the REFERENCE for the general forward of layers.attention.MultiHeadAttention (modules/layers/attention.py:125-176) -- the
cross-attention / n-dimensional / key-value-cache uses that FLAVA's encoder layers do not exercise:
python -m tests.golden.make_golden_mha_general  ->  mha_general.npz
  cross.*   dim_q 128, dim_kv 192, 2 heads: q [2,5,128] attends kv [2,9,192] under a key-padding mask [2,1,1,9]; output + probabilities
  grid.*    self-attention over a [2,3,4,128] token grid (n-dimensional input, :46-57) with a [12,12] mask
  dec.*     causal decoding with the cache (:159-176): a 4-token prefix, then two single-token steps (use_cache=True, causal=True),
            against the full 6-token pass under a causal mask; the cache tensors after the last step
  mem.*     non-causal cache: cross-attention to a fixed memory -- the second call reuses the cached keys / values (kv ignored)
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
    from torchmultimodal.modules.layers.attention import MultiHeadAttention, SelfAttention

    torch.set_num_threads(8)
    st = {}
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        seed(51)
        cross = MultiHeadAttention(dim_q=128, dim_kv=192, n_head=2, attn_module=SelfAttention()).eval()
        q, kv = torch.randn(2, 5, 128, generator=g), torch.randn(2, 9, 192, generator=g)
        km = (torch.rand(2, 1, 1, 9, generator=g) > 0.3).long()
        km[..., 0] = 1
        y, p = cross(q, kv, return_attn_weights=True, attention_mask=km)
        st.update({"cross.q": tnp(q), "cross.kv": tnp(kv), "cross.mask": tnp(km), "cross.out": tnp(y), "cross.probs": tnp(p)})
        st.update({"cross.sd." + k: v for k, v in sd_np(cross).items()})

        seed(52)
        selfa = MultiHeadAttention(dim_q=128, dim_kv=128, n_head=2, attn_module=SelfAttention()).eval()
        xg = torch.randn(2, 3, 4, 128, generator=g)
        gm = (torch.rand(12, 12, generator=g) > 0.2).long()
        gm[:, 0] = 1
        yg, pg = selfa(xg, return_attn_weights=True, attention_mask=gm)
        st.update({"grid.x": tnp(xg), "grid.mask": tnp(gm), "grid.out": tnp(yg), "grid.probs": tnp(pg)})
        st.update({"grid.sd." + k: v for k, v in sd_np(selfa).items()})

        # causal decoding: full pass vs prefix + steps through the cache
        x = torch.randn(2, 6, 128, generator=g)
        causal = torch.ones(6, 6).tril()
        full = selfa(x, attention_mask=causal)
        selfa.cache = None
        o0 = selfa(x[:, :4], use_cache=True, causal=True, attention_mask=torch.ones(4, 4).tril())
        o1 = selfa(x[:, 4:5], use_cache=True, causal=True)
        o2 = selfa(x[:, 5:6], use_cache=True, causal=True)
        st.update({"dec.x": tnp(x), "dec.full": tnp(full), "dec.o0": tnp(o0), "dec.o1": tnp(o1), "dec.o2": tnp(o2),
                   "dec.cache_k": tnp(selfa.cache["k"]), "dec.cache_v": tnp(selfa.cache["v"])})
        selfa.cache = None

        # fixed memory: the cache replaces kv on the second call
        cross.cache = None
        m0 = cross(q, kv, use_cache=True)
        q2 = torch.randn(2, 3, 128, generator=g)
        m1 = cross(q2, torch.zeros(2, 9, 192), use_cache=True)  # kv is ignored: keys / values come from the cache
        st.update({"mem.q2": tnp(q2), "mem.o0": tnp(m0), "mem.o1": tnp(m1), "mem.cache_k": tnp(cross.cache["k"])})
    np.savez_compressed(OUT / "mha_general.npz", **st)
    print("mha_general.npz", {k: v.shape for k, v in st.items() if not k.split(".")[1] == "sd"})
    assert np.abs(st["dec.full"][:, 4] - st["dec.o1"][:, 0]).max() < 1e-5 and np.abs(st["dec.full"][:, 5] - st["dec.o2"][:, 0]).max() < 1e-5


if __name__ == "__main__":
    main()
