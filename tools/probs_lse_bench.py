"""A/B of mmamd_attention_probs_fwd without a key mask (FLAVA's image / multimodal encoders): the r05 default (flash forward that parks the
log-sum-exp in the probability tensor + the one-pass whole-line probabilities kernel, csrc/attention_probs_lse.hip) against the two-pass
kernel (mmamd_debug_set_attn_variant(514)), alternating in one process, plus the parts of the new path on their own and a plain fill of
the same bytes.      python tools/probs_lse_bench.py [--shapes 256x197x12,128x275x12,256x77x12]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402


def timed(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="256x197x12,128x275x12,256x77x12,256x192x12,256x208x12")
    args = ap.parse_args()
    L = _lib.lib()
    for shp in args.shapes.split(","):
        B, S, H = (int(v) for v in shp.split("x"))
        torch.manual_seed(0)
        qkv = torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda()
        out = torch.empty((B * S, H * 64), dtype=torch.bfloat16, device="cuda")
        fill = torch.empty((B, H, S, S), dtype=torch.float32, device="cuda")
        res = {}
        for rnd in range(2):  # alternate the arms (clock / power state drifts over a process)
            for name, code in (("two_pass", 514), ("flash+one_pass", 515)):
                L.mmamd_debug_set_attn_variant(code)
                before = {k: ops.launch_count(k) for k in ("attention_probs_fwd", "attention_probs_lse")}
                res.setdefault(name, []).append(timed(lambda: ops.attention_probs_fwd(qkv, B, S, H, None, True, torch.float32, out=out)))
                ran = {k: ops.launch_count(k) - v for k, v in before.items()}
                new_path = S >= 112 and ops.attention_probs_from_lse_supported(S)
                # the arm ran the kernel it is named after (and only that one): an A/B of a knob that did not take effect is not printed
                assert (ran["attention_probs_lse"] > 0) == (code == 515 and new_path) and (ran["attention_probs_fwd"] > 0) == (code == 514 or not new_path), (name, ran)
                res.setdefault(name + " (no probs)", []).append(timed(lambda: ops.attention_probs_fwd(qkv, B, S, H, None, False, out=out)))
        L.mmamd_debug_set_attn_variant(515)
        res["flash forward + lse alone"] = [timed(lambda: ops.attention_fwd_train(qkv, B, S, H, False))]
        res["fill_ of the probability bytes"] = [timed(lambda: fill.fill_(0.5))]
        # the new path's values against the two-pass kernel's
        L.mmamd_debug_set_attn_variant(514)
        o0, p0 = ops.attention_probs_fwd(qkv, B, S, H, None, True, torch.float32)
        L.mmamd_debug_set_attn_variant(515)
        o1, p1 = ops.attention_probs_fwd(qkv, B, S, H, None, True, torch.float32)
        dp, do = (p0 - p1).abs().max().item(), (o0.float() - o1.float()).abs().max().item()
        mb = B * H * S * S * 4 / 1e6
        print(f"B={B} S={S} H={H}: probabilities {mb:.0f} MB; new vs two-pass max |dP| {dp:.2e}, max |dO| {do:.2e}")
        for k, v in res.items():
            print(f"    {k:34s} " + "  ".join(f"{t:7.1f} us" for t in v) + (f"   ({mb / min(v):.2f} TB/s of probability bytes)" if "no probs" not in k and "flash forward" not in k else ""))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
