#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace (csv) and reports how much kernels of different queues overlapped in time:
   tools/stream_overlap.py DIR   (DIR holds *_kernel_trace.csv)"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"][:60]) for r in rows]
ev.sort()
t0 = ev[0][0]
busy = sum(e - s for s, e, _, _ in ev)
span = max(e for _, e, _, _ in ev) - t0
pts = sorted([(s, 1) for s, _, _, _ in ev] + [(e, -1) for _, e, _, _ in ev])
depth = 0; last = pts[0][0]; hist = {}
for t, d in pts:
    hist[depth] = hist.get(depth, 0) + t - last
    last = t; depth += d
print("queues:", sorted(set(q for _, _, q, _ in ev)), " kernels:", len(ev))
print("span %.2f ms, sum of kernel durations %.2f ms" % (span / 1e6, busy / 1e6))
print("time with k kernels in flight (ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
tail = ev[-400:]
for s, e, q, n in tail[:24]:
    print("  q%s %9.1f -> %9.1f us  %s" % (q, (s - t0) / 1e3, (e - t0) / 1e3, n))
