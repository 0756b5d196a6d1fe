"""Step-boundary probe for rocprofv3 traces: python tools/boundary_probe.py MODE [steps]
MODE  plain: model(inp, finalize=False) back to back;  event: + a torch event record per step;  ahead: pipeline.RunAhead (the bench loop)"""
import gc
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from casmtr_amd.pipeline import HotPath, HotPathConfig, RunAhead, make_synthetic_inputs

mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
cfg = HotPathConfig.named("4c")
dev = torch.device("cuda", 0)
model = HotPath(cfg).to(dev)
inp = make_synthetic_inputs(cfg, 8, dev, seed=1234)
with torch.no_grad():
    model.qta.weight.copy_(inp["weight"])
    for _ in range(3):
        model(inp)
    torch.cuda.synchronize()
    gc.collect(); gc.freeze(); gc.disable()
    if mode == "ahead":
        ra = RunAhead(model)
        for _ in range(n):
            ra.submit(inp)
        ra.drain()
    else:
        keep = []
        for _ in range(n):
            keep.append(model(inp, finalize=False))
            if mode == "event":
                ev = torch.cuda.Event(); ev.record()
            if len(keep) > 2:
                keep.pop(0)
    torch.cuda.synchronize()
