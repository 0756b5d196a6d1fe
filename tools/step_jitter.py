"""Per-step wall time over many steps, with allocator / GC counters, to find where step-time spikes come from."""
import gc, sys, time
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd.pipeline import HotPath, HotPathConfig, make_synthetic_inputs

cfg = HotPathConfig(callers="--with-callers" in sys.argv)
dev = torch.device("cuda", 0)
model = HotPath(cfg).to(dev)
inp = make_synthetic_inputs(cfg, 8, dev, seed=1)
for _ in range(3):
    model(inp)
torch.cuda.synchronize()


def run(n, label):
    ts, notes = [], []
    for i in range(n):
        s0 = torch.cuda.memory_stats()
        g0 = gc.get_count()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model(inp)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        s1 = torch.cuda.memory_stats()
        ts.append((t2 - t0) * 1e3)
        if ts[-1] > 24:
            notes.append((i, round(ts[-1], 1), "enq %.1f" % ((t1 - t0) * 1e3),
                          "segs +%d" % (s1["segment.all.allocated"] - s0["segment.all.allocated"]),
                          "frees +%d" % (s1["segment.all.freed"] - s0["segment.all.freed"]),
                          "retries +%d" % (s1["num_alloc_retries"] - s0["num_alloc_retries"]), "gc", g0))
        del out
    ts_s = sorted(ts)
    print(f"{label}: median {ts_s[len(ts)//2]:.2f} mean {sum(ts)/len(ts):.2f} max {ts_s[-1]:.2f}  spikes: {notes}")


run(40, "default")
gc.disable()
run(40, "gc disabled")
gc.enable()
print("reserved GB", torch.cuda.memory_reserved() / 1e9, "allocated GB", torch.cuda.memory_allocated() / 1e9)
