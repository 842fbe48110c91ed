"""Where the time of a step goes, from a rocprofv3 kernel trace (`rocprofv3 --kernel-trace --output-format csv`): per queue the busy time, the time the
device runs nothing at all, the time exactly one / two queues are busy, and the kernels in front of the longest idle gaps.
    python tools/step_timeline.py <kernel_trace.csv> [--skip-ms 4000] [--top 12]
(--skip-ms: ignore everything before that many ms after the first kernel: start-up, warm-up steps.)"""
import argparse
import csv
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--last-ms", type=float, default=300.0, help="analyse the last so many ms of the trace (steady-state steps)")
    ap.add_argument("--top", type=int, default=12)
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", r.get("Stream_Id", "0")), r["Kernel_Name"]))
    rows.sort()
    t_end = max(r[1] for r in rows)
    t0 = t_end - int(a.last_ms * 1e6)
    rows = [r for r in rows if r[0] >= t0]
    span = (t_end - rows[0][0]) / 1e6
    per_q = defaultdict(float)
    ev = []
    for s, e, q, _ in rows:
        per_q[q] += (e - s) / 1e6
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    depth, last, hist = 0, ev[0][0], defaultdict(float)
    for t, d in ev:
        hist[min(depth, 3)] += (t - last) / 1e6
        depth += d
        last = t
    print(f"window {span:.1f} ms, {len(rows)} kernels; kernel time per queue: " + ", ".join(f"q{q}: {v:.1f} ms" for q, v in sorted(per_q.items())))
    print("device runs 0 / 1 / 2 / 3+ kernels at once: " + " / ".join(f"{hist[k]:.1f} ms ({100 * hist[k] / span:.1f} %)" for k in range(4)))
    # idle gaps: intervals with depth 0, and what ran right before / after
    gaps, depth, last_end_name = [], 0, ""
    evn = sorted([(s, 1, n) for s, e, q, n in rows] + [(e, -1, n) for s, e, q, n in rows])
    idle_from = None
    for t, d, n in evn:
        if d == 1:
            if depth == 0 and idle_from is not None:
                gaps.append((t - idle_from[0], idle_from[1], n))
            depth += 1
        else:
            depth -= 1
            if depth == 0:
                idle_from = (t, n)
    gaps.sort(reverse=True)
    tot = sum(g[0] for g in gaps) / 1e6
    print(f"{len(gaps)} idle gaps, {tot:.2f} ms in all; over 20 us: {sum(1 for g in gaps if g[0] > 20000)} ({sum(g[0] for g in gaps if g[0] > 20000) / 1e6:.2f} ms); the longest:")
    for g, before, after in gaps[:a.top]:
        print(f"   {g / 1e3:8.1f} us   after {before[:70]}   before {after[:70]}")
    by_before = defaultdict(lambda: [0, 0.0])
    for g, before, after in gaps:
        k = before.split("(")[0][:80]
        by_before[k][0] += 1; by_before[k][1] += g / 1e6
    print("idle time by the kernel that ran before the gap:")
    for k, (n, ms) in sorted(by_before.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print(f"   {ms:7.2f} ms in {n:5d} gaps   {k}")


if __name__ == "__main__":
    main()
