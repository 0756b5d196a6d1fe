"""Idle time between consecutive kernels of the launch stream, from a rocprofv3 kernel trace of bench.py:
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- python bench.py --steps 30 --warmup 5 --no-extra --no-cpu-baseline
  python tools/trace_gaps.py <dir>/*/*_kernel_trace.csv [steps]
The launch stream is the one that carries the fine-level kernels; steps are delimited by the dual-softmax GEMM."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = re.sub(r"<.*", "", r["Kernel_Name"].replace("void ", ""))[:28]
main = next(r for r in rows if "fine_quad" in r["n"])
ms = sorted((r for r in rows if (r["Queue_Id"], r["Stream_Id"]) == (main["Queue_Id"], main["Stream_Id"])), key=lambda r: r["s"])
gi = [i for i, r in enumerate(ms) if "ds_gemm16" in r["Kernel_Name"]]
seg = ms[gi[-nsteps - 1]:gi[-1]]
span, ksum = seg[-1]["e"] - seg[0]["s"], sum(r["e"] - r["s"] for r in seg)
print(f"{nsteps} steps: span {span / nsteps / 1e6:.3f} ms/step, kernel sum {ksum / nsteps / 1e6:.3f} ms/step, {len(seg) / nsteps:.1f} launches/step")
gap = collections.defaultdict(lambda: [0, 0])
for x, y in zip(seg, seg[1:]):
    g = gap[(x["n"], y["n"])]
    g[0] += y["s"] - x["e"]
    g[1] += 1
print(f"gaps {sum(v[0] for v in gap.values()) / nsteps / 1e6:.3f} ms/step")
for k, v in sorted(gap.items(), key=lambda kv: -kv[1][0])[:8]:
    print(f"{v[0] / nsteps / 1e3:7.1f} us/step {v[0] / v[1] / 1e3:7.1f} us avg x{v[1] / nsteps:5.1f}  {k[0]} -> {k[1]}")
dur = collections.defaultdict(lambda: [0, 0])
for r in seg:
    dur[r["n"]][0] += r["e"] - r["s"]
    dur[r["n"]][1] += 1
for k, v in sorted(dur.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"{v[0] / nsteps / 1e6:7.3f} ms/step x{v[1] / nsteps:5.1f} avg {v[0] / v[1] / 1e3:7.1f} us  {k}")
