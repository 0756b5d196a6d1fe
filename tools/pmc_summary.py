#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc CSV (counter_collection.csv) per kernel: mean counter value per dispatch."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
with open(path) as f:
    for row in csv.DictReader(f):
        k = row["Kernel_Name"].split("(")[0][:60]
        c = row["Counter_Name"]
        acc[k][c] += float(row["Counter_Value"])
        cnt[k][c] += 1
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    print(k, {c: round(acc[k][c] / cnt[k][c], 1) for c in acc[k]}, "n=", max(cnt[k].values()))
