#!/usr/bin/env python
"""What the chip does while the headline step's kernels run: socket power, power cap, shader / memory clocks and the
throttle / violation status, sampled by a side thread while each kernel (and the whole step) runs back to back for a few seconds.

    python tools/power_clocks.py [--seconds 5] [--hz 50] [--out gpurun_out/r05_power_clocks.json]

Telemetry sources, all tried, everything found is recorded (none of them is a derived counter ratio):
  * amdsmi (Python binding of libamd_smi, the library behind `amd-smi metric`): amdsmi_get_gpu_metrics_info (the firmware's gpu_metrics table:
    current_socket_power, current_gfxclks[8], current_uclk, average_gfx_activity, throttle_status, accumulated throttler residencies),
    amdsmi_get_power_info, amdsmi_get_power_cap_info, amdsmi_get_clock_info(GFX / MEM), amdsmi_get_violation_status;
  * sysfs hwmon of the card (power1_average / power1_input / power1_cap in microwatts, freq1_input / freq2_input in Hz);
  * one `amd-smi metric --json` / `amd-smi static --limit --json` text dump before and after, kept verbatim.
Workloads = the launches of one ViT-B/16 + text layer at B = 256 exactly as `CLIP.forward` issues them (grouped two-tower launches), each
captured in a HIP graph of `reps` launches and replayed for `--seconds`, then the whole step (bench.py's), then idle.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def _plain(v, depth=0):
    """amdsmi returns nested dicts / lists / enums: keep numbers and short strings."""
    if isinstance(v, (int, float, str, bool)) or v is None:
        return v
    if isinstance(v, dict) and depth < 4:
        return {str(k): _plain(x, depth + 1) for k, x in v.items()}
    if isinstance(v, (list, tuple)) and depth < 4:
        return [_plain(x, depth + 1) for x in v[:16]]
    return str(v)


class Telemetry:
    """Side-thread sampler.  `mark(label)` switches the label attached to the samples that follow."""

    METRIC_KEYS = ("current_socket_power", "average_socket_power", "current_gfxclk", "current_gfxclks", "average_gfxclk_frequency",
                   "current_uclk", "average_uclk_frequency", "current_socclk", "average_gfx_activity", "average_umc_activity",
                   "throttle_status", "indep_throttle_status", "temperature_hotspot", "temperature_mem", "accumulation_counter",
                   "prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc",
                   "gfxclk_lock_status", "firmware_timestamp", "system_clock_counter", "gfx_below_host_limit_acc", "gfx_below_host_limit_ppt_acc",
                   "gfx_below_host_limit_thm_acc", "gfx_below_host_limit_total_acc", "gfx_low_utilization_acc", "energy_accumulator", "gfx_activity_acc",
                   "mem_activity_acc", "temperature_vrsoc")

    def __init__(self, hz: float):
        self.dt = 1.0 / hz
        self.label = "start"
        self.samples = []
        self.errors = {}
        self.static = {}
        self._stop = threading.Event()
        self.smi, self.handle = None, None
        try:
            import amdsmi

            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self.smi, self.handle = amdsmi, hs[0]
            self.static["amdsmi_devices"] = len(hs)
        except Exception as e:  # noqa: BLE001
            self.errors["amdsmi_init"] = f"{type(e).__name__}: {e}"
        self.hwmon = None
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            if any(os.path.exists(os.path.join(d, f)) for f in ("power1_average", "power1_input", "freq1_input")):
                self.hwmon = d
                break
        self.static["hwmon"] = self.hwmon
        self._static_once()

    def _call(self, name, *a):
        try:
            return _plain(getattr(self.smi, name)(self.handle, *a))
        except Exception as e:  # noqa: BLE001
            self.errors.setdefault(name, f"{type(e).__name__}: {e}")
            return None

    def _static_once(self):
        if self.smi is not None:
            self.static["power_cap_info"] = self._call("amdsmi_get_power_cap_info")
            self.static["asic_info"] = self._call("amdsmi_get_gpu_asic_info")
            self.static["metrics_header"] = self._call("amdsmi_get_gpu_metrics_header_info")
            for dom in ("GFX", "MEM"):
                try:
                    self.static[f"clock_info_{dom}"] = self._call("amdsmi_get_clock_info", getattr(self.smi.AmdSmiClkType, dom))
                except Exception as e:  # noqa: BLE001
                    self.errors.setdefault("clk_enum", str(e))
        if self.hwmon:
            for f in ("power1_cap", "power1_cap_max", "power1_cap_default", "power1_label", "freq1_label", "freq2_label"):
                self.static["hwmon_" + f] = self._read(os.path.join(self.hwmon, f))
        for cmd in (["amd-smi", "static", "--limit", "--json"], ["amd-smi", "metric", "--power", "--clock", "--json"], ["amd-smi", "version"]):
            self.static["cli:" + " ".join(cmd[1:])] = self._cli(cmd)

    @staticmethod
    def _cli(cmd):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=30)
            return (r.stdout or r.stderr)[-6000:]
        except Exception as e:  # noqa: BLE001
            return f"{type(e).__name__}: {e}"

    @staticmethod
    def _read(path):
        try:
            return open(path).read().strip()
        except Exception:  # noqa: BLE001
            return None

    def sample(self):
        s = {"t": time.perf_counter(), "label": self.label}
        self._tick = getattr(self, "_tick", 0) + 1
        if self.smi is not None:
            m = self._call("amdsmi_get_gpu_metrics_info")
            if isinstance(m, dict):
                s["metrics"] = {k: m[k] for k in self.METRIC_KEYS if k in m}
        if self.smi is not None and self._tick % 4 == 1:  # the slow calls (violation status ~50 ms) on every 4th tick only
            p = self._call("amdsmi_get_power_info")
            if p is not None:
                s["power_info"] = p
            v = self._call("amdsmi_get_violation_status")
            if v is not None:
                s["violation"] = v
            try:
                s["clk_gfx"] = self._call("amdsmi_get_clock_info", self.smi.AmdSmiClkType.GFX)
                s["clk_mem"] = self._call("amdsmi_get_clock_info", self.smi.AmdSmiClkType.MEM)
            except Exception:  # noqa: BLE001
                pass
        if self.hwmon:
            for f in ("power1_average", "power1_input", "freq1_input", "freq2_input"):
                v = self._read(os.path.join(self.hwmon, f))
                if v is not None:
                    s["hwmon_" + f] = v
        self.samples.append(s)

    def _run(self):
        while not self._stop.is_set():
            t0 = time.perf_counter()
            self.sample()
            rest = self.dt - (time.perf_counter() - t0)
            if rest > 0:
                self._stop.wait(rest)

    def start(self):
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def stop(self):
        self._stop.set()
        self.th.join()
        self.static["cli_after:metric"] = self._cli(["amd-smi", "metric", "--power", "--clock", "--json"])

    def mark(self, label):
        self.label = label


def _num(v):
    try:
        return float(v)
    except Exception:  # noqa: BLE001
        return None


def summarise(samples, label, skip_s=1.0):
    """Per-workload summary over the samples after the first `skip_s` seconds of the workload (the power controller needs ~a second)."""
    ss = [s for s in samples if s["label"] == label]
    if not ss:
        return None
    t0 = ss[0]["t"]
    ss = [s for s in ss if s["t"] - t0 >= skip_s] or ss
    out = {"samples": len(ss)}

    def series(get):
        v = [x for x in (get(s) for s in ss) if x is not None]
        return v

    def stat(name, v, scale=1.0):
        if v:
            v = sorted(x * scale for x in v)
            out[name] = {"mean": round(sum(v) / len(v), 1), "min": round(v[0], 1), "p50": round(v[len(v) // 2], 1), "max": round(v[-1], 1)}

    stat("socket_power_W(metrics.current_socket_power)", series(lambda s: _num(s.get("metrics", {}).get("current_socket_power"))))
    stat("socket_power_W(power_info)", series(lambda s: _num((s.get("power_info") or {}).get("socket_power") or (s.get("power_info") or {}).get("current_socket_power"))))
    stat("power_W(hwmon power1_average)", series(lambda s: _num(s.get("hwmon_power1_average"))), 1e-6)
    stat("power_W(hwmon power1_input)", series(lambda s: _num(s.get("hwmon_power1_input"))), 1e-6)

    def gfx_mean(s):
        g = s.get("metrics", {}).get("current_gfxclks")
        if isinstance(g, list):
            g = [x for x in (_num(y) for y in g) if x is not None and 0 < x < 60000]
            return sum(g) / len(g) if g else None
        return _num(s.get("metrics", {}).get("current_gfxclk"))

    def gfx_min(s):
        g = s.get("metrics", {}).get("current_gfxclks")
        if isinstance(g, list):
            g = [x for x in (_num(y) for y in g) if x is not None and 0 < x < 60000]
            return min(g) if g else None
        return None

    stat("gfxclk_MHz(metrics.current_gfxclks mean over XCDs)", series(gfx_mean))
    stat("gfxclk_MHz(metrics.current_gfxclks min over XCDs)", series(gfx_min))
    stat("gfxclk_MHz(clock_info GFX clk)", series(lambda s: _num((s.get("clk_gfx") or {}).get("clk") or (s.get("clk_gfx") or {}).get("cur_clk"))))
    stat("sclk_MHz(hwmon freq1_input)", series(lambda s: _num(s.get("hwmon_freq1_input"))), 1e-6)
    stat("uclk_MHz(metrics.current_uclk)", series(lambda s: _num(s.get("metrics", {}).get("current_uclk"))))
    stat("mclk_MHz(hwmon freq2_input)", series(lambda s: _num(s.get("hwmon_freq2_input"))), 1e-6)
    stat("gfx_activity_pct", series(lambda s: _num(s.get("metrics", {}).get("average_gfx_activity"))))
    stat("umc_activity_pct", series(lambda s: _num(s.get("metrics", {}).get("average_umc_activity"))))
    stat("temp_hotspot_C", series(lambda s: _num(s.get("metrics", {}).get("temperature_hotspot"))))
    # residency accumulators: the delta over the workload / the delta of the accumulation counter = the fraction of time the limiter was active
    first, last = ss[0].get("metrics", {}), ss[-1].get("metrics", {})
    acc0, acc1 = _num(first.get("accumulation_counter")), _num(last.get("accumulation_counter"))
    for k in ("ppt_residency_acc", "prochot_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc",
              "gfx_below_host_limit_acc", "gfx_below_host_limit_ppt_acc", "gfx_below_host_limit_thm_acc", "gfx_below_host_limit_total_acc",
              "gfx_low_utilization_acc"):
        a, b = first.get(k), last.get(k)
        if isinstance(a, list) and isinstance(b, list):
            d = [(_num(y) or 0) - (_num(x) or 0) for x, y in zip(a, b)]
            out[k + "_delta"] = d[:8]
        elif _num(a) is not None and _num(b) is not None:
            out[k + "_delta"] = _num(b) - _num(a)
    if acc0 is not None and acc1 is not None:
        out["accumulation_counter_delta"] = acc1 - acc0
    ts = {str(s.get("metrics", {}).get("throttle_status")) for s in ss} | {"indep:" + str(s.get("metrics", {}).get("indep_throttle_status")) for s in ss}
    out["throttle_status_values_seen"] = sorted(ts)[:8]
    vs = [s["violation"] for s in ss if isinstance(s.get("violation"), dict)]
    v0, v1 = (vs[0], vs[-1]) if len(vs) >= 2 else (None, vs[-1] if vs else None)
    if isinstance(v1, dict):
        out["violation_last"] = {k: v for k, v in v1.items() if isinstance(v, (int, float, str, bool))}
        if isinstance(v0, dict):
            out["violation_acc_delta"] = {k: _num(v1[k]) - _num(v0[k]) for k in v1 if k.startswith("acc_") and _num(v1.get(k)) is not None and _num(v0.get(k)) is not None}
            # r06 -- WHY the shader clock sits below the host limit, per XCD of partition 0 (firmware residency counters, same time base as acc_counter):
            # _pwr = the socket power limit (PPT), _thm = a thermal limit, _total = any reason; total - pwr - thm = a limiter the counters do not name
            # (electrical: peak-current / voltage-droop management of the matrix pipes).  acc_low_utilization = clock lowered because the XCD idles.
            dc = (_num(v1.get("acc_counter")) or 0) - (_num(v0.get("acc_counter")) or 0)

            def xcd_delta(key):
                a, b = v0.get(key), v1.get(key)
                if isinstance(a, list) and isinstance(b, list) and a and isinstance(a[0], list):
                    d = [(_num(y), _num(x)) for x, y in zip(a[0], b[0])]
                    d = [y - x for y, x in d if x is not None and y is not None]
                    return d
                return None

            if dc > 0:
                rep = {"acc_counter_delta": dc}
                for key in ("acc_gfx_clk_below_host_limit_pwr", "acc_gfx_clk_below_host_limit_thm", "acc_gfx_clk_below_host_limit_total", "acc_low_utilization"):
                    d = xcd_delta(key)
                    if d:
                        rep[key + "_share_mean_over_xcds"] = round(sum(d) / len(d) / dc, 4)
                        rep[key + "_share_per_xcd"] = [round(x / dc, 3) for x in d]
                t_, p_, h_ = (rep.get("acc_gfx_clk_below_host_limit_%s_share_mean_over_xcds" % k) for k in ("total", "pwr", "thm"))
                if t_ is not None and p_ is not None and h_ is not None:
                    rep["below_host_limit_unnamed_reason_share"] = round(t_ - p_ - h_, 4)
                out["clock_limit_reasons"] = rep
    # energy_accumulator (firmware's accumulated socket energy; 15.259 uJ per count on this family): mean power over the workload without sampling noise
    es = [(s["t"], _num(s.get("metrics", {}).get("energy_accumulator"))) for s in ss]
    es = [(t, e) for t, e in es if e is not None]
    if len(es) >= 2 and es[-1][0] > es[0][0]:
        out["energy_accumulator_delta"] = es[-1][1] - es[0][1]
        out["mean_power_W(energy_accumulator x 15.259 uJ / wall time)"] = round((es[-1][1] - es[0][1]) * 15.259e-6 / (es[-1][0] - es[0][0]), 1)
    return out


def external(args):
    """--cmd mode: telemetry around external commands (microbenchmarks), one summary per command."""
    tel = Telemetry(args.hz)
    tel.start()
    time.sleep(1.5)
    outs = {}
    for spec in args.cmd:
        label, _, cmd = spec.partition("=")
        tel.mark(label)
        r = subprocess.run(cmd, shell=True, capture_output=True, text=True)
        tel.mark("gap")
        outs[label] = {"cmd": cmd, "rc": r.returncode, "stdout": r.stdout[-2000:], "stderr": r.stderr[-500:]}
        time.sleep(1.0)
    tel.stop()
    rec = {"what": "power / clock telemetry around external commands (tools/power_clocks.py --cmd)", "hz_requested": args.hz, "static": {k: v for k, v in tel.static.items() if not k.startswith("cli")},
           "errors": tel.errors, "workloads": {}}
    for label, o in outs.items():
        t = summarise(tel.samples, label, 1.0)
        rec["workloads"][label] = {"run": o, "telemetry": t}
        pw = (t or {}).get("socket_power_W(metrics.current_socket_power)")
        ck = (t or {}).get("gfxclk_MHz(metrics.current_gfxclks mean over XCDs)")
        cr = (t or {}).get("clock_limit_reasons") or {}
        print(f"{label:28s} {o['stdout'].strip()[-200:]}\n{'':28s} power {pw} (energy counter: {(t or {}).get('mean_power_W(energy_accumulator x 15.259 uJ / wall time)')} W) gfxclk {ck} ppt {(t or {}).get('ppt_residency_acc_delta')} / {(t or {}).get('accumulation_counter_delta')}"
              f"\n{'':28s} clock below host limit: any reason {cr.get('acc_gfx_clk_below_host_limit_total_share_mean_over_xcds')}, power {cr.get('acc_gfx_clk_below_host_limit_pwr_share_mean_over_xcds')}, "
              f"thermal {cr.get('acc_gfx_clk_below_host_limit_thm_share_mean_over_xcds')}, unnamed {cr.get('below_host_limit_unnamed_reason_share')}, low utilisation {cr.get('acc_low_utilization_share_mean_over_xcds')}")
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(rec, indent=1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=5.0)
    ap.add_argument("--hz", type=float, default=50.0)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "r05_power_clocks.json"))
    ap.add_argument("--raw", action="store_true", help="also keep every sample in the file")
    ap.add_argument("--cmd", action="append", default=[], help="LABEL=COMMAND: sample while this external command runs (repeatable) instead of the built-in workloads")
    args = ap.parse_args()
    if args.cmd:
        return external(args)

    import torch

    from multimodal_amd import build, ops

    build.build()
    dev = torch.device("cuda", 0)
    tel = Telemetry(args.hz)
    tel.start()
    B = args.batch
    Mv, Mt = B * 197, B * 77
    bf, f32 = torch.bfloat16, torch.float32

    def rnd(*shape, dtype=bf, scale=1.0):
        return (torch.randn(*shape, device=dev) * scale).to(dtype)

    def gemm_pair(Nv, Kv, Nt, Kt, act, res):
        od = f32 if res else bf
        av, wv, bv = rnd(Mv, Kv), rnd(Nv, Kv, scale=0.05), rnd(Nv, dtype=f32)
        at, wt, bt = rnd(Mt, Kt), rnd(Nt, Kt, scale=0.05), rnd(Nt, dtype=f32)
        ov, ot = torch.zeros(Mv, Nv, dtype=od, device=dev), torch.zeros(Mt, Nt, dtype=od, device=dev)
        fl = 2.0 * Mv * Nv * Kv + 2.0 * Mt * Nt * Kt
        return (lambda: ops.gemm_bf16_grouped([(av, wv, bv, ov if res else None, ov), (at, wt, bt, ot if res else None, ot)], act=act, out_dtype=od)), fl

    work = {}
    work["gemm_qkv(grouped ViT 2304x768 + text 1536x512)"] = gemm_pair(2304, 768, 1536, 512, ops.ACT_NONE, False)
    work["gemm_out_proj+fp32_residual(grouped 768x768 + 512x512)"] = gemm_pair(768, 768, 512, 512, ops.ACT_NONE, True)
    work["gemm_mlp_up+quickgelu(grouped 3072x768 + 2048x512)"] = gemm_pair(3072, 768, 2048, 512, ops.ACT_QUICKGELU, False)
    work["gemm_mlp_down+fp32_residual(grouped 768x3072 + 512x2048)"] = gemm_pair(768, 3072, 512, 2048, ops.ACT_NONE, True)
    qv, qt = rnd(Mv, 2304), rnd(Mt, 1536)
    ov_, ot_ = torch.empty(Mv, 768, dtype=bf, device=dev), torch.empty(Mt, 512, dtype=bf, device=dev)
    work["attention(grouped ViT 3072x[197,64] + text 2048x[77,64] causal)"] = (
        lambda: ops.attention_fwd_grouped([(qv, B, 197, 12, False, ov_), (qt, B, 77, 8, True, ot_)]), 4.0 * B * 12 * 197 * 197 * 64 + 4.0 * B * 8 * 77 * 77 * 64)
    xv, xt = rnd(Mv, 768, dtype=f32), rnd(Mt, 512, dtype=f32)
    gv, bv_, gt, bt_ = torch.ones(768, device=dev), torch.zeros(768, device=dev), torch.ones(512, device=dev), torch.zeros(512, device=dev)
    yv, yt = torch.empty(Mv, 768, dtype=bf, device=dev), torch.empty(Mt, 512, dtype=bf, device=dev)
    work["layernorm(grouped ViT + text rows, fp32 -> bf16)"] = (lambda: ops.add_layernorm_grouped([(xv, None, gv, bv_, 1e-5, yv), (xt, None, gt, bt_, 1e-5, yt)]), 0.0)

    results = {}
    st = torch.cuda.Stream(device=dev)

    def run_graph(label, fn, flops, reps=20):
        with torch.cuda.stream(st):
            for _ in range(3):
                fn()
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(reps):
                    fn()
            g.replay()
            st.synchronize()
            tel.mark(label)
            t0 = time.perf_counter()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 0
            e0.record(st)
            while time.perf_counter() - t0 < args.seconds:
                for _ in range(8):
                    g.replay()
                n += 8
                st.synchronize() if n % 64 == 0 else None
            e1.record(st)
            st.synchronize()
            tel.mark("gap")
        us = e0.elapsed_time(e1) * 1e3 / (n * reps)
        r = {"launch_us": round(us, 2), "launches": n * reps}
        if flops:
            r["TFLOPs"] = round(flops / us / 1e6, 1)
            r["mfma_frac_of_2500"] = round(flops / us / 1e6 / 2500.0, 4)
        results[label] = r
        time.sleep(0.5)

    time.sleep(2.0)
    tel.mark("idle_before")
    time.sleep(2.0)
    tel.mark("gap")
    for label, (fn, fl) in work.items():
        run_graph(label, fn, fl)

    # the whole step (bench.py's): both towers + loss, eager launches on the current stream
    from multimodal_amd.models.clip import clip_vit_b16
    from multimodal_amd.modules.losses.contrastive_loss_with_temperature import ContrastiveLossWithTemperature
    from multimodal_amd.utils.synthetic import clip_batch

    torch.manual_seed(0)
    model = clip_vit_b16().to(dev).eval()
    loss_fn = ContrastiveLossWithTemperature().to(dev)
    images, ids = clip_batch(B)
    images_d, ids_d = images.to(dev), ids.to(dev)
    with torch.no_grad():
        for _ in range(5):
            out = model(images_d, ids_d)
            loss_fn(out.embeddings_a, out.embeddings_b)
        torch.cuda.synchronize()
        label = "whole_step(CLIP ViT-B/16 B=256 forward + loss, eager)"
        tel.mark(label)
        t0 = time.perf_counter()
        n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.perf_counter() - t0 < max(args.seconds, 8.0):
            out = model(images_d, ids_d)
            loss_fn(out.embeddings_a, out.embeddings_b)
            n += 1
            if n % 16 == 0:
                torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        tel.mark("gap")
        ms = e0.elapsed_time(e1) / n
        results[label] = {"ms_per_step": round(ms, 3), "steps": n, "mfma_frac_of_2500": round(B / ms * 1e3 * 41.09e9 / 2.5e15, 4)}
    time.sleep(1.0)
    tel.mark("idle_after")
    time.sleep(2.0)
    tel.stop()

    rec = {"what": "power / clock telemetry per kernel of the headline step (tools/power_clocks.py); first second of each workload skipped in the summaries",
           "hz_requested": args.hz, "seconds_per_workload": args.seconds, "static": tel.static, "errors": tel.errors, "workloads": {}}
    for label in ["idle_before"] + list(results) + ["idle_after"]:
        rec["workloads"][label] = {"timing": results.get(label), "telemetry": summarise(tel.samples, label, 0.0 if label.startswith("idle") else 1.0)}
    if args.raw:
        rec["raw"] = tel.samples
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(rec, indent=1))
    # compact table on stdout
    print(f"telemetry sources: amdsmi={'yes' if tel.smi else 'no'} hwmon={tel.hwmon}; errors: {list(tel.errors)[:6]}")
    for label, w in rec["workloads"].items():
        t = w["telemetry"] or {}
        pw = next((t[k] for k in t if k.startswith("socket_power_W") or k.startswith("power_W")), None)
        ck = next((t[k] for k in t if k.startswith("gfxclk_MHz") or k.startswith("sclk_MHz")), None)
        cr = t.get("clock_limit_reasons") or {}
        print(f"{label[:64]:64s} {json.dumps(w['timing'])}  power {pw}  gfxclk {ck}  ppt_delta {t.get('ppt_residency_acc_delta')} / acc {t.get('accumulation_counter_delta')}  "
              f"energy-counter W {t.get('mean_power_W(energy_accumulator x 15.259 uJ / wall time)')}  below-limit any/pwr/thm/unnamed/lowutil "
              f"{cr.get('acc_gfx_clk_below_host_limit_total_share_mean_over_xcds')}/{cr.get('acc_gfx_clk_below_host_limit_pwr_share_mean_over_xcds')}/"
              f"{cr.get('acc_gfx_clk_below_host_limit_thm_share_mean_over_xcds')}/{cr.get('below_host_limit_unnamed_reason_share')}/{cr.get('acc_low_utilization_share_mean_over_xcds')}")


if __name__ == "__main__":
    main()
