"""Weight-gradient GEMM dW = dY^T X: row-major operands (TN kernel) vs transposes + NT split-K.  python tools/wgrad_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

for name, T, M, N in (("v.qkv", 50432, 2304, 768), ("v.out", 50432, 768, 768), ("v.up", 50432, 3072, 768), ("v.down", 50432, 768, 3072),
                      ("t.qkv", 19712, 1536, 512), ("t.up", 19712, 2048, 512)):
    y = (torch.randn(T, M) * 0.1).to(torch.bfloat16).cuda()
    x = torch.randn(T, N).to(torch.bfloat16).cuda()
    tn = timeit(lambda: ops.gemm_bf16_tn_splitk(y, x), 20) * 1e3
    if "--sched" in sys.argv:  # experiment builds only (MMAMD_EXPERIMENTS=1): fragment-read placement variants of the TN loop
        alt = []
        for v in (40, 41, 42, 43, 44, 45, 46, 47):
            ops.set_gemm_variant(v)
            alt.append(timeit(lambda: ops.gemm_bf16_tn_splitk(y, x), 20) * 1e3)
        ops.set_gemm_variant(0)
        print(f"{name:7s} TN policy {tn:7.1f} us | MFMA-first {alt[0]:7.1f} | burst {alt[1]:7.1f} | compiler {alt[2]:7.1f} | reads-first {alt[3]:7.1f} | GM=1 {alt[4]:7.1f} | GM=2 {alt[5]:7.1f} | flat GM=1 {alt[6]:7.1f} | flat GM=8 {alt[7]:7.1f}", flush=True)
        continue

    def old():
        yt, _ = ops.transpose_to_bf16(y, pad_to=128, with_colsum=True)
        xt = ops.transpose_to_bf16(x, pad_to=128)
        return ops.gemm_bf16_splitk(yt, xt)
    nt = timeit(old, 20) * 1e3
    yt, _ = ops.transpose_to_bf16(y, pad_to=128, with_colsum=True)
    xt = ops.transpose_to_bf16(x, pad_to=128)
    nt_only = timeit(lambda: ops.gemm_bf16_splitk(yt, xt), 20) * 1e3
    cs = timeit(lambda: ops.colsum(y), 20) * 1e3
    fl = 2.0 * T * M * N
    print(f"{name:7s} T={T} M={M} N={N}:  TN {tn:7.1f} us ({fl / tn / 1e6:6.1f} TF/s)   NT gemm only {nt_only:7.1f} us   transposes+NT {nt:7.1f} us   colsum(dY) {cs:6.1f} us", flush=True)
