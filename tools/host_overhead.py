"""Host-side cost of one hot-path step: enqueue time (no sync) vs GPU time, with / without the HIP-event hooks and the
match gather, + optional cProfile of the enqueue (--cprofile)."""
import cProfile, pstats, sys, time, io
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import _lib, dist as cdist
from casmtr_amd.pipeline import HotPath, HotPathConfig, make_synthetic_inputs

callers = "--with-callers" in sys.argv
cfg = HotPathConfig(callers=callers)
dev = torch.device("cuda", 0)
model = HotPath(cfg).to(dev)
inp = make_synthetic_inputs(cfg, 8, dev, seed=1)
for _ in range(3):
    model(inp)
torch.cuda.synchronize()
for prof, gather in ((False, False), (True, False), (False, True), (True, True), (False, False)):
    _lib.prof_enable(prof)
    ts = []
    for rep in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model(inp)
        if gather:
            cdist.gather_matches(out, pairs_per_rank=8)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    _lib.prof_enable(False)
    print(f"prof={prof} gather={gather}: step ms {['%.2f' % t for t in ts]}")
if "--cprofile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    out = model(inp)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
    print(s.getvalue()[:6000])
