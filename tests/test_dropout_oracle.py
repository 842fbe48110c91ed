"""oracle/philox.py against the Random123 known-answer vectors of philox4x32-10 (kat_vectors: the three philox4x32 10-round lines) and the
statistical / structural properties the dropout masks rely on.  CPU-only."""
import numpy as np

from oracle import philox


def test_philox4x32_10_known_answers():
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox.philox4x32_10(*[np.array([c], dtype=np.uint32) for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want


def test_mask_structure_and_rate():
    n, p = 1 << 20, 0.3
    m = philox.dropout_mask(n, p, seed=1234, site=5)
    assert m.dtype == np.uint8 and m.shape == (n,) and set(np.unique(m)) <= {0, 1}
    assert abs(m.mean() - (1 - p)) < 4 * np.sqrt(p * (1 - p) / n)          # binomial 4-sigma
    assert not np.array_equal(m, philox.dropout_mask(n, p, seed=1234, site=6))  # another site: another stream
    assert not np.array_equal(m, philox.dropout_mask(n, p, seed=1235, site=5))
    assert np.array_equal(m, philox.dropout_mask(n, p, seed=1234, site=5))      # a pure function of (seed, site, index)
    assert np.array_equal(m[:1000], philox.dropout_mask(1000, p, seed=1234, site=5))  # prefix property (counter = element group)
    assert philox.dropout_mask(64, 0.0, 1, 0).all()                             # p = 0 keeps everything
    g = philox.dropout_mask(8 * 12, 0.5, seed=7, site=1, group=12)              # stochastic depth: one decision per sample
    assert all(len(set(g[s * 12:(s + 1) * 12])) == 1 for s in range(8))


def test_apply_scaling():
    x = np.arange(-8, 8, dtype=np.float32)
    m = philox.dropout_mask(16, 0.25, 3, 0)
    y = philox.dropout_apply(x, m, 0.25)
    assert np.array_equal(y[m == 0], np.zeros((m == 0).sum(), np.float32))
    assert np.allclose(y[m == 1], x[m == 1] / 0.75, rtol=3e-7)
