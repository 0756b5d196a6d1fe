"""One stream, step k+1 enqueued before step k's match counts are read back (no idle gap at the host sync): pairs/s against the plain loop"""
import gc, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from casmtr_amd.pipeline import HotPath, HotPathConfig, make_synthetic_inputs

cfg = HotPathConfig.named("4c")
dev = torch.device("cuda", 0)
model = HotPath(cfg).to(dev)
inp = make_synthetic_inputs(cfg, 8, dev, seed=1234)
with torch.no_grad():
    model.qta.weight.copy_(inp["weight"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for _ in range(5):
    model(inp)
gc.collect(); gc.freeze(); gc.disable()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = model(inp)
    torch.cuda.synchronize(); t_plain = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pend = model(inp, finalize=False)
    for i in range(n):
        nxt = model(inp, finalize=False) if i + 1 < n else None
        out = model.finalize(pend)
        pend = nxt
    torch.cuda.synchronize(); t_def = time.perf_counter() - t0
    print(f"plain {8 * n / t_plain:.1f} pairs/s ({t_plain / n * 1e3:.3f} ms)   deferred finalize {8 * n / t_def:.1f} pairs/s ({t_def / n * 1e3:.3f} ms)")
