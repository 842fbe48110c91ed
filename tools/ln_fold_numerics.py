"""Numerics study for folding LayerNorm into the consumer GEMM ("levers not yet pulled").  CPU only, numpy.

  standard   y = bf16(LN(x)) . bf16(W)^T + b                      (today: layernorm kernel -> GEMM)
  folded     y = rstd * (bf16(x) . bf16(g*W)^T - mu * colsum(bf16(g*W))) + (beta . W^T + b)
             (x rounded to bf16 by the producing epilogue; mu, rstd from fp32 sum / sum-of-squares partials)

Both accumulate in fp32 (emulated in float64 on bf16-rounded operands: the accumulation error is far below the operand rounding).
Reports the error of each against the exact float64 result, relative to the RMS of y, as a function of the per-token mean / std ratio
of the residual stream and of a few outlier channels ("massive activations").

    python tools/ln_fold_numerics.py
"""
import numpy as np


def bf16(a):
    a = np.asarray(a, np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).astype(np.float64)


def study(ratio, outliers, rng, rows=512, d=768, n=256):
    x = rng.standard_normal((rows, d))
    if outliers:
        idx = rng.choice(d, outliers, replace=False)
        x[:, idx] *= 40.0                      # a few channels carry most of the variance
    x = x / x.std(1, keepdims=True)
    x = (x + ratio) * rng.uniform(0.5, 20.0, (rows, 1))   # per-token mean = ratio * std; token scales vary
    x = x.astype(np.float32).astype(np.float64)          # the fp32 residual stream
    g = 1.0 + 0.1 * rng.standard_normal(d)
    beta = 0.05 * rng.standard_normal(d)
    W = 0.02 * rng.standard_normal((n, d))
    b = 0.01 * rng.standard_normal(n)
    mu = x.mean(1, keepdims=True)
    var = (x * x).mean(1, keepdims=True) - mu * mu        # one-pass statistics, as epilogue partials would give
    rstd = 1.0 / np.sqrt(var + 1e-5)
    ln = (x - mu) * rstd * g + beta
    exact = ln @ W.T + b
    std = bf16(ln) @ bf16(W).T + b
    Wg = bf16(W * g)
    fold = rstd * (bf16(x) @ Wg.T - mu * Wg.sum(1)) + (beta @ W.T + b)
    scale = np.sqrt((exact ** 2).mean())
    e_std = np.abs(std - exact)
    e_fold = np.abs(fold - exact)
    return e_std.max() / scale, np.sqrt((e_std ** 2).mean()) / scale, e_fold.max() / scale, np.sqrt((e_fold ** 2).mean()) / scale


def main():
    rng = np.random.default_rng(0)
    print(f"{'mean/std':>8s} {'outliers':>8s} | {'standard max':>12s} {'rms':>9s} | {'folded max':>12s} {'rms':>9s} | rms ratio")
    for outliers in (0, 4):
        for ratio in (0.0, 0.25, 0.5, 1.0, 2.0, 5.0, 10.0):
            a, b, c, d = study(ratio, outliers, rng)
            print(f"{ratio:8.2f} {outliers:8d} | {a:12.2e} {b:9.2e} | {c:12.2e} {d:9.2e} | {d / b:6.2f}")


if __name__ == "__main__":
    main()
