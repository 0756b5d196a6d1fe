"""Whole HotPath step as a HIP graph (casmtr_amd/graph.py) against the eager loop: python tools/graph_probe.py [steps] [config]
Also checks that the per-kernel HIP events recorded during capture can be read after the replays (bench.py's roofline numbers)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import _lib
from casmtr_amd.graph import GraphedHotPath
from casmtr_amd.pipeline import HotPath, HotPathConfig, make_synthetic_inputs

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
cfg = HotPathConfig.named(sys.argv[2] if len(sys.argv) > 2 else "4c")
dev = torch.device("cuda:0")
model = HotPath(cfg).to(dev)
inp = make_synthetic_inputs(cfg, 8, dev, seed=1234)
with torch.no_grad():
    model.qta.weight.copy_(inp["weight"])
for _ in range(5):
    out = model(inp)
ref = {k: out[k].clone() for k in ("m_bids", "mkpts0", "mkpts1", "mconf")}


def run(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


eager = run(lambda: model(inp), steps)
print(f"eager, read-back every step: {eager:.3f} ms/step ({8e3 / eager:.1f} pairs/s), {ref['m_bids'].numel()} matches")
if os.environ.get("PROBE_PROF", "1") == "1":
    _lib.prof_enable_only("qta_fine_level[lists<=64]")
gs = GraphedHotPath(model, inp)
g_ms = run(gs.step, steps)
o = gs.step()
same = all(torch.equal(o[k], ref[k]) for k in ref)
print(f"graph, read-back every step: {g_ms:.3f} ms/step ({8e3 / g_ms:.1f} pairs/s), outputs identical to eager: {same}")
mode = os.environ.get("PROBE_MODE", "ahead")
if mode == "none":
    sys.exit(0)
pend = {"o": None}


def ahead():
    new = gs.enqueue()
    old, pend["o"] = pend["o"], new
    if old is not None:
        gs.finalize(old)


def replay_only():      # back-to-back replays, nothing read
    gs.enqueue()


def ahead_event():      # one step ahead, but the host waits for the previous replay's end before launching the next graph
    ev = torch.cuda.Event()
    new = gs.enqueue()
    ev.record()
    old, pend["o"] = pend["o"], new
    if old is not None:
        gs.finalize(old)
    ev.synchronize()


fn = {"ahead": ahead, "replay": replay_only, "event": ahead_event}[mode]
a_ms = run(fn, steps)
if pend["o"] is not None:
    gs.finalize(pend["o"])
print(f"graph, mode {mode}: {a_ms:.3f} ms/step ({8e3 / a_ms:.1f} pairs/s)")
try:
    print("events of the last replays:", {k: (round(v[0], 3), v[1]) for k, v in _lib.prof_read().items()})
except RuntimeError as e:
    print("events recorded inside the captured graph cannot be read back:", e)
_lib.prof_enable(False)
