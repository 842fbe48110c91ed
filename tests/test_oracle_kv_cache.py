"""Pin the key/value-cache paths of the numpy oracle (mha_with_cache / layers_decoder_layer with past, use_cache) to the reference
(fixture tests/golden/kv_cache.npz from make_golden_kv_cache.py).  CPU-only."""
import numpy as np

from oracle import clip_oracle as oc


def _sd(z, prefix):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


def test_mha_with_cache_past_and_use_cache(golden):
    """reference modules/layers/multi_head_attention.py:158-179 (the shape of tests/modules/layers/test_multi_head_attention.py:127-140
    at kernel-legal sizes)."""
    z = golden("kv_cache.npz")
    sd = _sd(z, "mha.sd.")
    out, (k, v) = oc.mha_with_cache(z["mha.x"], z["mha.x"], sd, "", 2, attend=z["mha.mask"], past=(z["mha.pk"], z["mha.pv"]), use_cache=True)
    np.testing.assert_allclose(out, z["mha.out"], atol=2e-5)
    np.testing.assert_allclose(k, z["mha.key"], atol=1e-6)
    np.testing.assert_allclose(v, z["mha.value"], atol=1e-6)
    assert np.array_equal(k[:, :, :5], z["mha.pk"])
    out2, (k2, _) = oc.mha_with_cache(z["mha.x"], z["mha.x"], sd, "", 2, use_cache=True)
    np.testing.assert_allclose(out2, z["mha.out_nopast"], atol=2e-5)
    np.testing.assert_allclose(k2, z["mha.key_nopast"], atol=1e-6)


def test_decoder_incremental_decoding_equals_full_pass(golden):
    """TransformerDecoder with past_key_values / use_cache (modules/layers/transformer.py:586-657): a prefix of 4 positions, then one
    position at a time, reproduces the full causal pass and the reference's final caches."""
    z = golden("kv_cache.npz")
    sd = _sd(z, "dec.sd.")
    h, enc = z["dec.h"], z["dec.enc"]
    causal = np.tril(np.ones((6, 6), dtype=bool))

    def run(x, attend, caches):
        new = []
        for i in range(2):
            x, pr = oc.layers_decoder_layer(x, enc, sd, f"layer.{i}.", 2, 1e-5, attend, past=None if caches is None else caches[i], use_cache=True)
            new.append(pr)
        return oc.layer_norm(x, sd["final_layer_norm.weight"], sd["final_layer_norm.bias"], 1e-5), new

    y, cache = run(h[:, :4], causal[:4, :4], None)
    outs = [y]
    for t in (4, 5):
        y, cache = run(h[:, t:t + 1], causal[t:t + 1, :t + 1], cache)
        outs.append(y)
    np.testing.assert_allclose(np.concatenate(outs, axis=1), z["dec.full"], atol=3e-5)
    np.testing.assert_allclose(cache[1][0], z["dec.cache_k1"], atol=2e-5)
    np.testing.assert_allclose(cache[0][1], z["dec.cache_v0"], atol=2e-5)
    np.testing.assert_allclose(oc.layers_decoder(h, enc, sd, "", 2, 1e-5, causal, final_eps=1e-5), z["dec.full"], atol=3e-5)
