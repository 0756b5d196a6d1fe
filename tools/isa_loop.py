#!/usr/bin/env python3
"""Instruction mix of the hottest loop of a kernel in a gfx950 assembly file: the span between a label and the LAST backward branch to
it with the most instructions in between.  usage: isa_loop.py file.s kernel-name-substring"""
import collections
import re
import sys

t = open(sys.argv[1]).read()
name = next(n for n in re.findall(r"^(_Z\w+):", t, re.M) if sys.argv[2] in n)
body = t.split(name + ":", 1)[1].split(".Lfunc_end", 1)[0].split("\n")
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"(\.LBB\w+):", l))}
best = (0, 0, 0)
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch\w*\s+(\.LBB\w+)|\s+s_branch\s+(\.LBB\w+)", l)
    if m:
        tgt = m.group(1) or m.group(2)
        if tgt in labels and labels[tgt] < i and i - labels[tgt] > best[0]:
            best = (i - labels[tgt], labels[tgt], i)
_, lo, hi = best
kind, ops = collections.Counter(), collections.Counter()
for l in body[lo:hi + 1]:
    code = l.split(";")[0]
    m = re.match(r"\s+([a-z_0-9]+)", code)
    if not m:
        continue
    op = m.group(1)
    k = ("mfma" if "mfma" in op else "lds" if op.startswith("ds_") else "valu" if op.startswith("v_") else
         "salu" if op.startswith("s_") else "vmem" if op.startswith(("global", "buffer", "flat")) else "other")
    kind[k] += 1
    ops[op] += 1
print(name, "loop lines", lo, "-", hi, dict(kind))
print(ops.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25))
