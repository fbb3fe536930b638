#!/usr/bin/env python3
"""Warm-only per-kernel statistics from a rocprofv3 --kernel-trace directory: the first SKIP calls of every kernel (the
bench's warm-up steps, whose launches include cold instruction caches and clock ramp) are dropped, so that
`roofline.achieved` (live HIP events over the timed steps) and flop / AverageNs from this file describe the same launches.

usage: kstats_warm.py DIR OUT.csv [SKIP=2]      (run the bench with only its main leg: --no-vit --no-bf16 --no-upload
                                                 --no-directory --no-cpu-baseline, so every call of a kernel is the same shape)"""
import csv, glob, os, sys

d, out = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 2
calls = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        calls.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows = []
for name, ts in calls.items():
    ts.sort()
    warm = [e - s for s, e in ts[skip:]] if len(ts) > skip else []
    if not warm:
        continue
    rows.append((sum(warm), name, len(warm), sum(warm) / len(warm), min(warm), max(warm), len(ts)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows) or 1
with open(out, "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "CallsIncludingSkipped", "SkippedFirst"])
    for t, name, n, avg, mn, mx, nall in rows:
        w.writerow([name, n, t, f"{avg:.1f}", f"{100.0 * t / tot:.2f}", mn, mx, nall, skip])
for t, name, n, avg, mn, mx, nall in rows[:14]:
    print(f"{name[:90]:90s} calls={n:4d} avg_us={avg / 1e3:10.1f} min_us={mn / 1e3:10.1f} max_us={mx / 1e3:10.1f}")
