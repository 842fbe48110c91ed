"""The four weight gradients of a layer: one split-K launch + reduce each (the shipped form) against ONE grouped launch + one reduce
(mmamd_gemm_bf16_tn_splitk_group) at several split counts.    python tools/wgrad_group_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

for name, T, probs in (("ViT-B/16 layer, 50432 tokens", 50432, [(768, 3072, True), (3072, 768, True), (768, 768, False), (2304, 768, True)]),
                       ("text layer, 19712 tokens", 19712, [(512, 2048, True), (2048, 512, True), (512, 512, False), (1536, 512, True)]),
                       ("ViT-L/14 layer, 32896 tokens", 32896, [(1024, 4096, True), (4096, 1024, True), (1024, 1024, False), (3072, 1024, True)])):
    jobs = [((torch.randn(T, M) * 0.1).to(torch.bfloat16).cuda(), torch.randn(T, N).to(torch.bfloat16).cuda(), cs) for M, N, cs in probs]
    fl = sum(2.0 * T * M * N for M, N, _ in probs)

    def single():
        return [ops.gemm_bf16_tn_splitk(y, x, want_colsum=True) if cs else ops.gemm_bf16_tn_splitk(y, x) for y, x, cs in jobs]
    each = [timeit((lambda y=y, x=x, cs=cs: ops.gemm_bf16_tn_splitk(y, x, want_colsum=cs)) if cs else (lambda y=y, x=x: ops.gemm_bf16_tn_splitk(y, x)), 20) * 1e3
            for y, x, cs in jobs]
    t1 = timeit(single, 20) * 1e3
    tiles = sum(((M + 255) // 256) * ((N + 255) // 256) for M, N, _ in probs)
    auto = ops.wgrad_group_splits(tiles, T // 64)
    print(f"{name}: {fl / 1e9:.0f} GF, {tiles} tiles; one launch + reduce per Linear {t1:7.1f} us ({fl / t1 / 1e6:6.1f} TF/s)  [" + " ".join(f"{t:.0f}" for t in each) + f"]; auto splits {auto}", flush=True)
    for s in sorted({1, 2, 3, 4, 5, 7, 9, 12, 14, auto}):
        if T // 64 // s < 16:
            continue
        before = ops.launch_count("gemm_bf16_tn_splitk_group")
        tg = timeit(lambda: ops.gemm_bf16_tn_splitk_group(jobs, splits=s), 20) * 1e3
        assert ops.launch_count("gemm_bf16_tn_splitk_group") > before
        print(f"    grouped, splits {s:2d}: {tiles * s:5d} workgroups ({tiles * s / 256:5.2f} rounds)  {tg:7.1f} us ({fl / tg / 1e6:6.1f} TF/s)", flush=True)
