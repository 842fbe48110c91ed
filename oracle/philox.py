"""TEST INFRASTRUCTURE (oracle): numpy restatement of the counter-based generator behind the training-time dropout masks of the MI355X path
(multimodal_amd/csrc/dropout.hip).  Only tests/ may import this.

Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 `philox4x32_R(10, ...)`), pinned below to the
Random123 known-answer vectors (tests/test_dropout_oracle.py).  The reference (torch's nn.Dropout / torchvision's StochasticDepth:
modules/layers/mlp.py:59-60, modules/layers/transformer.py:64-70) draws its masks from torch's generator; a different stream is a different,
equally valid sample, so parity is stated per mask: with the mask this module predicts, forward and gradients must equal the reference
arithmetic applied with that same mask.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
U32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays (counters) with scalar or array keys.  Returns the 4 output words."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32).copy() for c in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = (p1 & U32).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = (p0 & U32).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def threshold(p: float) -> int:
    """keep <=> r >= threshold(p) on a 32-bit output r (dropout.hip: integer compare, P(keep) = 1 - threshold / 2^32)."""
    t = float(np.float32(p)) * 4294967296.0
    return 0xFFFFFFFF if t >= 4294967295.0 else int(t)


def dropout_mask(n: int, p: float, seed: int, site: int, group: int = 0) -> np.ndarray:
    """uint8 [n] keep mask of mmamd_dropout: group = 0 -> one decision per element (counter = (i // 4, site), output word i % 4);
    group > 0 -> one decision per sample of `group` elements (sample s: counter (s // 4, site), output word s % 4)."""
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    th = np.uint32(threshold(p))
    if group == 0:
        idx = np.arange((n + 3) // 4, dtype=np.uint64)
    else:
        ns = n // group
        idx = np.arange((ns + 3) // 4, dtype=np.uint64)
    lo, hi = (idx & U32).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32)
    words = np.stack(philox4x32_10(lo, hi, np.full_like(lo, site), np.zeros_like(lo), k0, k1), axis=1).reshape(-1)
    if group == 0:
        return (words[:n] >= th).astype(np.uint8)
    return np.repeat((words[:n // group] >= th).astype(np.uint8), group)


def dropout_apply(x: np.ndarray, mask: np.ndarray, p: float) -> np.ndarray:
    """x * mask / (1 - p) in float32 with the kernel's operation order (scale = 1 / (1 - p) rounded to fp32, one multiply)."""
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    return np.where(mask.reshape(x.shape) != 0, x.astype(np.float32) * scale, np.float32(0.0)).astype(np.float32)
