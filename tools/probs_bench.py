"""attention_probs_fwd timing (FLAVA shapes): with / without the probability output.  python tools/probs_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402


def serial(on: bool) -> None:  # A/B switch of the probability kernel's key loops (mmamd_debug_set_attn_variant 512 / 513)
    _lib.lib().mmamd_debug_set_attn_variant(512 if on else 513)


for B, S, H in ((128, 197, 12), (128, 77, 12), (128, 275, 12)):
    qkv = torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda()
    for dt in (torch.float32, torch.bfloat16):
        res = {}
        for mode in ("serial", "pipelined"):
            serial(mode == "serial")
            o, p = ops.attention_probs_fwd(qkv, B, S, H, None, want_probs=True, probs_dtype=dt)
            t1 = timeit(lambda: ops.attention_probs_fwd(qkv, B, S, H, None, want_probs=True, probs_dtype=dt), 30) * 1e3
            t0 = timeit(lambda: ops.attention_probs_fwd(qkv, B, S, H, None, want_probs=False), 30) * 1e3
            res[mode] = (t1, t0, o, p)
        serial(False)
        same = torch.equal(res["serial"][2], res["pipelined"][2]) and torch.equal(res["serial"][3], res["pipelined"][3])
        mb = B * H * S * S * (4 if dt == torch.float32 else 2) / 1e6
        print(f"B={B} S={S} H={H} probs {str(dt)[6:]:8s} ({mb:.0f} MB): serial {res['serial'][0]:7.1f} us with / {res['serial'][1]:7.1f} without | "
              f"pipelined {res['pipelined'][0]:7.1f} with / {res['pipelined'][1]:7.1f} without | {'==' if same else '!='}", flush=True)
