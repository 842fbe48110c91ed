#!/usr/bin/env python
"""A/B of the fp32-residual GEMM epilogue across two BUILDS of the library (r06: pipelined buffer loads / stores against the r01-r05 form):
alternating child processes, one per build (MMAMD_LIB), each timing the grouped out-projection and MLP-down launches of the cfg-2 layer (both
towers) and the ViT-only launches, and printing a digest of the outputs on fixed inputs -- the two builds must agree bit for bit.

    python tools/resid_epilogue_ab.py --base multimodal_amd/lib_base/libmmamd_r05.so [--rounds 3]
    python tools/resid_epilogue_ab.py --child          (one build: the library MMAMD_LIB names, or the in-tree one)"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def child(batch):
    import torch

    from multimodal_amd import ops
    from tools.kernel_bench import timeit

    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    if os.environ.get("MMAMD_AB_STAGGER"):
        from multimodal_amd import _lib

        _lib.lib().mmamd_debug_set_gemm_stagger(int(os.environ["MMAMD_AB_STAGGER"]))

    def rnd(*shape, dtype=torch.bfloat16, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev).to(dtype)

    Mv, Mt = batch * 197, batch * 77
    out = {}
    for name, pv, pt in (("out_proj", (Mv, 768, 768), (Mt, 512, 512)), ("mlp_down", (Mv, 768, 3072), (Mt, 512, 2048))):
        def prob(M, N, K):
            x = rnd(M, N, dtype=torch.float32)
            return [rnd(M, K), rnd(N, K, scale=0.05), rnd(N, dtype=torch.float32), x, x]  # in place: x += A W^T + bias

        v, t = prob(*pv), prob(*pt)
        # digest on the fixed inputs: one grouped launch, one ViT-only launch into a separate output, ragged M (edge rows), residual != output
        ops.gemm_bf16_grouped([tuple(v), tuple(t)], act=0, out_dtype=torch.float32)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        h.update(v[4].cpu().numpy().tobytes()); h.update(t[4].cpu().numpy().tobytes())
        Me = Mv - 37
        r2 = rnd(Me, pv[1], dtype=torch.float32)
        c2 = torch.full((Me, pv[1]), 7.0, dtype=torch.float32, device=dev)
        ops.gemm_bf16_grouped([(v[0][:Me], v[1], v[2], r2, c2), tuple(t)], act=0, out_dtype=torch.float32)
        torch.cuda.synchronize()
        h.update(c2.cpu().numpy().tobytes())
        out[name + "_digest"] = h.hexdigest()[:16]
        out[name + "_grouped_us"] = timeit(lambda: ops.gemm_bf16_grouped([tuple(v), tuple(t)], act=0, out_dtype=torch.float32), 10) * 1e3
        out[name + "_vit_only_us"] = timeit(lambda: ops.gemm_bf16_grouped([tuple(v)], act=0, out_dtype=torch.float32), 10) * 1e3
    # the bf16-output launches (qkv, MLP-up + QuickGELU): their stores go through buffer descriptors too (r06) and the next tile's first barrier
    # no longer drains them
    for name, pv, pt, act in (("qkv", (Mv, 2304, 768), (Mt, 1536, 512), 0), ("mlp_up", (Mv, 3072, 768), (Mt, 2048, 512), 1)):
        def prob16(M, N, K):
            return (rnd(M, K), rnd(N, K, scale=0.05), rnd(N, dtype=torch.float32), None, torch.zeros((M, N), dtype=torch.bfloat16, device=dev))

        v, t = prob16(*pv), prob16(*pt)
        ops.gemm_bf16_grouped([v, t], act=act, out_dtype=torch.bfloat16)
        Me = Mv - 37
        c2 = torch.full((Me, pv[1]), 7.0, dtype=torch.bfloat16, device=dev)
        ops.gemm_bf16_grouped([(v[0][:Me], v[1], v[2], None, c2)], act=act, out_dtype=torch.bfloat16)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for x in (v[4], t[4], c2):
            h.update(x.view(torch.int16).cpu().numpy().tobytes())
        out[name + "_digest"] = h.hexdigest()[:16]
        out[name + "_grouped_us"] = timeit(lambda: ops.gemm_bf16_grouped([v, t], act=act, out_dtype=torch.bfloat16), 10) * 1e3
        out[name + "_vit_only_us"] = timeit(lambda: ops.gemm_bf16_grouped([v], act=act, out_dtype=torch.bfloat16), 10) * 1e3
    print("AB_RESULT " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--base", default=str(ROOT / "multimodal_amd" / "lib_base" / "libmmamd_r05.so"))
    ap.add_argument("--extra", default="", help="more arms: name=lib.so[@stagger-per-cent],... (empty lib = the in-tree build)")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    if a.child:
        return child(a.batch)
    arms = [("base", a.base, None), ("new", None, None)]
    for spec in filter(None, a.extra.split(",")):  # name=lib.so[@stagger]   (lib empty = the in-tree build)
        name, _, rest = spec.partition("=")
        lib, _, stg = rest.partition("@")
        arms.append((name, lib or None, stg or None))
    res = {arm: [] for arm, _, _ in arms}
    for _ in range(a.rounds):
        for arm, lib, stg in arms:
            env = dict(os.environ)
            env.pop("MMAMD_LIB", None)
            env.pop("MMAMD_AB_STAGGER", None)
            if lib:
                env["MMAMD_LIB"] = lib
                env["MMAMD_LIB_ALLOW_MISSING"] = "1"
            if stg:
                env["MMAMD_AB_STAGGER"] = stg
            p = subprocess.run([sys.executable, __file__, "--child", "--batch", str(a.batch)], env=env, capture_output=True, text=True, timeout=900)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("AB_RESULT ")]
            if p.returncode != 0 or not line:
                raise SystemExit(f"{arm} failed:\n{p.stdout[-2000:]}\n{p.stderr[-2000:]}")
            res[arm].append(json.loads(line[-1][len("AB_RESULT "):]))
    keys = [k for k in res["base"][0] if k.endswith("_us")]
    print(f"fp32-residual GEMM epilogue, B = {a.batch}, {a.rounds} alternating rounds (median us per launch)")
    for k in keys:
        med = {arm: sorted(r[k] for r in res[arm])[a.rounds // 2] for arm in res}
        print(f"  {k:24s} " + "   ".join(f"{arm} {v:7.1f} ({(v / med['base'] - 1) * 100:+.1f} %)" for arm, v in med.items()))
        print(f"  {'':24s} all: " + "  ".join(f"{arm} {[round(r[k], 1) for r in res[arm]]}" for arm in res))
    for k in [k for k in res["base"][0] if k.endswith("_digest")]:
        ds = {arm: {r[k] for r in res[arm]} for arm in res}
        same = all(d == ds["base"] and len(d) == 1 for d in ds.values())
        print(f"  {k:24s} {'BIT-IDENTICAL across arms' if same else 'DIFFERENT: ' + str(ds)}")


if __name__ == "__main__":
    main()
