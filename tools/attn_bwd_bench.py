"""attention_bwd (dQ + dK/dV kernels) at the ViT-B/16 and text shapes of a B = 256 training step.   python tools/attn_bwd_bench.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from multimodal_amd import _lib, ops  # noqa: E402

L = _lib.lib()


def main():
    for (B, S, H, causal) in ((256, 197, 12, False), (256, 77, 8, True)):
        torch.manual_seed(0)
        qkv = torch.randn(B * S, 3 * H * 64).to(torch.bfloat16).cuda()
        dout = torch.randn(B * S, H * 64).to(torch.bfloat16).cuda()
        out, lse = ops.attention_fwd_train(qkv, B, S, H, causal)
        res = {}
        for rnd in range(2):
            for code, tag in ((4003, "default"), (4000, "two kernels (r03)"), (4001, "single pass"), (4002, "fused (two roles)")):
                L.mmamd_debug_set_attn_variant(code)
                for _ in range(2):
                    d = ops.attention_bwd(qkv, out, dout, lse, B, S, H, causal)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    d = ops.attention_bwd(qkv, out, dout, lse, B, S, H, causal)
                e1.record()
                torch.cuda.synchronize()
                res[code] = d
                print(f"S={S} causal={int(causal)} {tag}: {e0.elapsed_time(e1) * 200:7.1f} us", flush=True)
        L.mmamd_debug_set_attn_variant(4003)
        print(f"S={S}: fused == two kernels bit for bit: {torch.equal(res[4002], res[4000])}", flush=True)
        D = H * 64
        ref, got = res[4000].float(), res[4001].float()
        for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
            err = float((got[:, sl] - ref[:, sl]).abs().max())
            print(f"S={S}: single pass vs two kernels {name}: max |d| {err:.3e} of max |ref| {float(ref[:, sl].abs().max()):.3e}", flush=True)


if __name__ == "__main__":
    main()
