"""attention_probs_fwd timing (FLAVA shapes): with / without the probability output.  python tools/probs_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

for B, S, H in ((128, 197, 12), (128, 77, 12), (128, 275, 12)):
    qkv = torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda()
    for dt in (torch.float32, torch.bfloat16):
        t1 = timeit(lambda: ops.attention_probs_fwd(qkv, B, S, H, None, want_probs=True, probs_dtype=dt), 30) * 1e3
        t0 = timeit(lambda: ops.attention_probs_fwd(qkv, B, S, H, None, want_probs=False), 30) * 1e3
        mb = B * H * S * S * (4 if dt == torch.float32 else 2) / 1e6
        print(f"B={B} S={S} H={H} probs {str(dt)[6:]:8s}: {t1:7.1f} us with probabilities ({mb:.0f} MB -> {mb / (t1 - t0) / 1e3 if t1 > t0 else 0:.2f} TB/s marginal), {t0:7.1f} us without", flush=True)
