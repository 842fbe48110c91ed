#!/usr/bin/env python
"""Energy budget of the headline step: where the ~18 J of one CLIP ViT-B/16 B = 256 forward + loss go.

Inputs (both written by `bash tools/gpu_energy_parts.sh` on an MI355X):
  gpurun_out/r06_energy_parts.json   tools/microbench/energy_parts.hip under the telemetry sampler: ONE resource at a time (LDS reads, L2->LDS DMA,
                                     HBM reads, HBM copy, fp32 VALU, the matrix pipe on N(0,1) operands, matrix pipe + LDS, + DMA) -> rate, socket
                                     power (amdsmi current_socket_power and the firmware's energy accumulator), shader clock, limiter residencies
  gpurun_out/r06_power_clocks.json   the step's own launches and the whole step under the same sampler
Model (first order, additive): E_step = P_idle(clocks up) * t_step + sum over resources of units * marginal energy per unit, with
marginal energy per unit = (P_microbenchmark - P_idle) / rate.  The marginal energies are taken at each microbenchmark's own clock / voltage point;
the step runs its GEMMs at a LOWER clock (1.76-1.90 GHz against 2.37 GHz for the LDS / HBM / VALU loops), where a byte or a flop costs less -- the sum
is therefore an upper bound per line and the residual can be negative.  Units per step are structural counts (stated next to each line).

    python tools/energy_budget.py [--parts gpurun_out/r06_energy_parts.json] [--step gpurun_out/r06_power_clocks.json] [--out profiles/r06_energy_budget.json]
"""
import argparse
import json
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def power_of(t):
    e = (t or {}).get("mean_power_W(energy_accumulator x 15.259 uJ / wall time)")
    if e is not None:
        return float(e)
    return float(t["socket_power_W(metrics.current_socket_power)"]["mean"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", default=str(ROOT / "gpurun_out" / "r06_energy_parts.json"))
    ap.add_argument("--step", default=str(ROOT / "gpurun_out" / "r06_power_clocks.json"))
    ap.add_argument("--out", default=str(ROOT / "profiles" / "r06_energy_budget.json"))
    a = ap.parse_args()
    parts, step = json.loads(Path(a.parts).read_text()), json.loads(Path(a.step).read_text())
    p_idle = power_of(step["workloads"]["idle_before"]["telemetry"])  # clocks up (2.4 GHz), nothing running: the floor every kernel pays
    micro = {}
    for name, w in parts["workloads"].items():
        m = re.search(r"ENERGY_PART (\{.*\})", w["run"]["stdout"])
        if not m:
            continue
        r = json.loads(m.group(1))
        t = w["telemetry"]
        micro[name] = {"rate_G_per_s": r["rate"], "unit": r["unit"], "rate2_G_per_s": r.get("rate2"), "unit2": r.get("unit2"), "power_W": power_of(t),
                       "gfxclk_MHz": t["gfxclk_MHz(metrics.current_gfxclks mean over XCDs)"]["mean"],
                       "ppt_residency": round(t["ppt_residency_acc_delta"] / max(t["accumulation_counter_delta"], 1), 3),
                       "clock_limit_reasons": t.get("clock_limit_reasons")}
    pj = {k: (v["power_W"] - p_idle) / (v["rate_G_per_s"] * 1e9) * 1e12 for k, v in micro.items()}  # pJ per byte / per flop above the clocks-up idle
    ws = step["workloads"]
    whole = next(v for k, v in ws.items() if k.startswith("whole_step"))
    t_ms = whole["timing"]["ms_per_step"]
    p_step = power_of(whole["telemetry"])
    e_step = p_step * t_ms * 1e-3
    # ---- structural units of one step (CLIP ViT-B/16 + text, B = 256) ------------------------------------------------------------------------
    flops = 10.52e12
    gemm_flops = 12 * (209.5 + 69.8 + 279.3 + 279.3) * 1e9 + 2 * 256 * 196 * 768 * 768  # the four grouped projections of 12 layers + the stem
    lds_read_bytes = gemm_flops / 32768 * 768  # 6 ds_read_b128 (1 KiB each, wave-wide) per 8 MFMAs of 32 x 32 x 16 -> 768 B per MFMA
    dma_bytes = gemm_flops / 128               # a 256 x 256 tile DMAs 64 KiB per 64-deep K-tile = 2 * 256 * 256 * 64 flop -> flops / 128 bytes
    # HBM bytes per layer of both towers: PMC where measured (FETCH x 2 + WRITE, profiles/pmc_residual_kernel.json, r04_pmc_mlp_up_kernel.json,
    # r04_pmc_attention_kernel.json), algorithmic otherwise (qkv, LayerNorm, the text tower's attention)
    per_layer = {"qkv (algorithmic)": 348e6, "out-projection (PMC)": 541e6, "MLP-up (PMC)": 910e6, "MLP-down (PMC)": 1073e6, "attention (PMC + text algorithmic)": 391e6,
                 "2 x LayerNorm (algorithmic)": 586e6}
    hbm_bytes = 12 * sum(per_layer.values()) + 0.5e9  # + stem, heads, loss, image cast
    lines = [
        ("clocks-up idle floor (P_idle x t_step)", p_idle * t_ms * 1e-3, f"{p_idle:.0f} W x {t_ms:.2f} ms"),
        ("matrix pipe, N(0,1) operands", flops * pj["mfma"] * 1e-12, f"10.52 TFLOP x {pj['mfma']:.3f} pJ/flop (1826 TF/s at {micro['mfma']['power_W']:.0f} W, {micro['mfma']['gfxclk_MHz']:.0f} MHz)"),
        ("HBM traffic", hbm_bytes * 0.5 * (pj["hbm_read"] + pj["hbm_copy"]) * 1e-12,
         f"{hbm_bytes / 1e9:.1f} GB x {0.5 * (pj['hbm_read'] + pj['hbm_copy']):.0f} pJ/B (read {pj['hbm_read']:.0f}, copy {pj['hbm_copy']:.0f}; {micro['hbm_read']['rate_G_per_s'] / 1e3:.2f} / {micro['hbm_copy']['rate_G_per_s'] / 1e3:.2f} TB/s)"),
        ("LDS fragment reads of the GEMMs", lds_read_bytes * pj["lds_read"] * 1e-12, f"{lds_read_bytes / 1e9:.0f} GB x {pj['lds_read']:.1f} pJ/B ({micro['lds_read']['rate_G_per_s'] / 1e3:.0f} TB/s at {micro['lds_read']['power_W']:.0f} W)"),
        ("L2 -> LDS DMA of the GEMMs' operand tiles", dma_bytes * pj["l2_dma"] * 1e-12, f"{dma_bytes / 1e9:.0f} GB x {pj['l2_dma']:.1f} pJ/B ({micro['l2_dma']['rate_G_per_s'] / 1e3:.1f} TB/s at {micro['l2_dma']['power_W']:.0f} W)"),
    ]
    named = sum(x[1] for x in lines)
    lines.append(("everything else (VALU / SALU issue of epilogues, softmax, LayerNorm; L2; instruction fetch; model error)", e_step - named, "measured - the lines above"))
    out = {
        "what": "first-order energy budget of one headline step (CLIP ViT-B/16 B = 256 forward + loss) on one MI355X; tools/energy_budget.py",
        "step": {"ms_per_step": t_ms, "socket_power_W": round(p_step, 1), "energy_J": round(e_step, 2), "gfxclk_MHz": whole["telemetry"]["gfxclk_MHz(metrics.current_gfxclks mean over XCDs)"]["mean"],
                 "ppt_residency": round(whole["telemetry"]["ppt_residency_acc_delta"] / whole["telemetry"]["accumulation_counter_delta"], 3), "power_cap_W": 1400},
        "idle_clocks_up_W": round(p_idle, 1),
        "microbenchmarks": micro,
        "marginal_energy_pJ_per_unit": {k: round(v, 3) for k, v in pj.items()},
        "hbm_bytes_per_layer": per_layer, "hbm_bytes_per_step": hbm_bytes, "lds_read_bytes_per_step": lds_read_bytes, "l2_to_lds_dma_bytes_per_step": dma_bytes,
        "budget_J": [{"line": n, "J": round(j, 2), "share": round(j / e_step, 3), "how": how} for n, j, how in lines],
        "per_launch": {k: {"timing": v["timing"], "power_W": round(power_of(v["telemetry"]), 1), "gfxclk_MHz": v["telemetry"]["gfxclk_MHz(metrics.current_gfxclks mean over XCDs)"]["mean"],
                           "ppt_residency": round(v["telemetry"]["ppt_residency_acc_delta"] / max(v["telemetry"]["accumulation_counter_delta"], 1), 3),
                           "clock_limit_reasons": v["telemetry"].get("clock_limit_reasons")} for k, v in ws.items() if v.get("timing")},
        "reading": [
            "The matrix pipe's clock limit is NOT the socket power cap: the register-only MFMA loop on N(0,1) operands sits at ~1830 MHz at ~1290 W with PPT "
            "residency 2-31 % by run, while an LDS-read loop (1257 W) and an fp32 VALU loop (1248 W) at the SAME socket power hold 2371-2373 MHz.  None of "
            "the firmware's named limiters accounts for it (thermal / PROCHOT / VR / HBM residencies 0; gfx_clk_below_host_limit pwr / thm / total 0): the "
            "limiter is specific to dense MFMA issue on toggling operands (all-zero operands run 2380 MHz at 857 W, r05) and is not exposed through amdsmi "
            "on this firmware.  1826 TF/s = 0.73 of the nominal peak is what the pipe delivers on real data, whatever the kernel around it does.",
            "HBM traffic is the second-largest energy line after the matrix pipe: ~105 pJ per byte moved, ~47 GB per step.  The step is power-limited "
            "(PPT residency 0.74 on this box, 0.93 on r05's), so bytes are time even where no kernel is bandwidth-bound: 1 GB per step is ~0.1 J = ~0.08 ms at "
            "1366 W.  The over-fetch of the MLP-up launch (1.84 x algorithmic: 415 MB per layer) and the fp32 residual stream (0.93 GB per layer) are worth "
            "~0.4 ms and ~0.9 ms per step in energy terms -- the first is a target, the second is what parity is paid with.",
            "LDS fragment reads (9 %) and operand DMA (4 %) are small: consistent with r04's direct-W main loop (a third fewer LDS reads) moving neither time nor clock.",
        ],
    }
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text(json.dumps(out, indent=1) + "\n")
    print(f"step: {t_ms:.2f} ms at {p_step:.0f} W = {e_step:.2f} J   (idle, clocks up: {p_idle:.0f} W)")
    for n, j, how in lines:
        print(f"  {j:6.2f} J  {j / e_step * 100:5.1f} %  {n}   [{how}]")
    print("marginal energies (pJ per byte / flop above idle):", {k: round(v, 2) for k, v in pj.items()})


if __name__ == "__main__":
    main()
