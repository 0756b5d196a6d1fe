#!/usr/bin/env python3
"""Dump rocprofv3's (rocpd sqlite) kernel summary to CSV:  tools/rocpd_top_kernels.py results.db out.csv"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for r in cur:
        w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.3f}", f"{r[4]:.3f}"])
