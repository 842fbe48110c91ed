"""Key/value-cache fixtures from the REFERENCE (MultiHeadAttentionWithCache / TransformerDecoder with past_key_values, use_cache):
python -m tests.golden.make_golden_kv_cache  ->  kv_cache.npz
  mha.*   MultiHeadAttentionWithCache(128, 128, 2 heads): 3 new positions on top of 5 cached ones, boolean mask, use_cache=True
  dec.*   TransformerDecoder(2 pre-norm layers, d 128, 2 heads, cross-attention on 7 encoder states, GELU): full causal pass over 6
          positions vs incremental decoding (prefix of 4 with use_cache, then one position at a time with the returned caches)
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
    from torchmultimodal.modules.layers.multi_head_attention import MultiHeadAttentionWithCache
    from torchmultimodal.modules.layers.transformer import TransformerDecoder

    torch.set_num_threads(8)
    st = {}
    seed(41)
    mha = MultiHeadAttentionWithCache(dim_q=128, dim_kv=128, num_heads=2).eval()
    g = torch.Generator().manual_seed(3)
    B, Sp, Sn = 3, 5, 3
    x = torch.randn(B, Sn, 128, generator=g)
    pk, pv = torch.randn(B, 2, Sp, 64, generator=g), torch.randn(B, 2, Sp, 64, generator=g)
    mask = torch.rand(B, 1, Sn, Sp + Sn, generator=g) > 0.25
    mask[..., 0] = True
    with torch.no_grad():
        o = mha(x, x, x, attn_mask=mask, past_key_value=(pk, pv), use_cache=True)
        o2 = mha(x, x, x, use_cache=True)  # no past: the cache is just this call's keys / values
    st.update({"mha.x": tnp(x), "mha.pk": tnp(pk), "mha.pv": tnp(pv), "mha.mask": tnp(mask), "mha.out": tnp(o.attn_output),
               "mha.key": tnp(o.past_key_value[0]), "mha.value": tnp(o.past_key_value[1]), "mha.out_nopast": tnp(o2.attn_output),
               "mha.key_nopast": tnp(o2.past_key_value[0])})
    st.update({"mha.sd." + k: v for k, v in sd_np(mha).items()})

    seed(42)
    dec = TransformerDecoder(n_layer=2, d_model=128, n_head=2, dim_feedforward=256, activation=torch.nn.GELU, layer_norm_eps=1e-5,
                             norm_first=True, use_cross_attention=True, dim_kv=128, final_layer_norm_eps=1e-5).eval()
    B, S, Se = 2, 6, 7
    h = torch.randn(B, S, 128, generator=g)
    enc = torch.randn(B, Se, 128, generator=g)
    causal = torch.ones(S, S, dtype=torch.bool).tril()
    with torch.no_grad():
        full = dec(h, enc, attention_mask=causal)
        o = dec(h[:, :4], enc, attention_mask=causal[:4, :4], use_cache=True)
        steps = [tnp(o.last_hidden_state)]
        cache = o.current_key_values
        for t in (4, 5):
            o = dec(h[:, t:t + 1], enc, attention_mask=causal[t:t + 1, :t + 1], past_key_values=cache, use_cache=True)
            steps.append(tnp(o.last_hidden_state))
            cache = o.current_key_values
    inc = np.concatenate(steps, axis=1)
    assert np.abs(inc - tnp(full.last_hidden_state)).max() < 1e-5  # incremental == full in the reference
    st.update({"dec.h": tnp(h), "dec.enc": tnp(enc), "dec.full": tnp(full.last_hidden_state), "dec.cache_k1": tnp(cache[1][0]),
               "dec.cache_v0": tnp(cache[0][1])})
    st.update({"dec.sd." + k: v for k, v in sd_np(dec).items()})
    np.savez_compressed(OUT / "kv_cache.npz", **st)
    print({k: v.shape for k, v in st.items() if ".sd." not in k}, (OUT / "kv_cache.npz").stat().st_size)


if __name__ == "__main__":
    main()
