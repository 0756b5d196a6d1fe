#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel trace over its LAST `--window-ms` milliseconds (one steady-state step of a looped workload, leaving
out library auto-tuning during warm-up).  usage: last_step_stats.py <kernel_trace.csv> --window-ms 217 [--top 40]"""
import argparse
import collections
import csv

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--window-ms", type=float, required=True)
ap.add_argument("--top", type=int, default=40)
a = ap.parse_args()
rows = list(csv.DictReader(open(a.trace)))
end = max(int(r["End_Timestamp"]) for r in rows)
lo = end - int(a.window_ms * 1e6)
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= lo:
        k = agg[r["Kernel_Name"]]
        k[0] += 1
        k[1] += e - s
tot = sum(v[1] for v in agg.values())
print(f"name,calls,total_ms,percent   # window {a.window_ms} ms, kernel time {tot / 1e6:.2f} ms")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[: a.top]:
    print(f"\"{n[:150]}\",{c},{t / 1e6:.3f},{100 * t / tot:.2f}")
