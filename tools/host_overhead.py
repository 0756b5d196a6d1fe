#!/usr/bin/env python3
"""How long the host needs to enqueue one hot-path step (no device sync inside), next to the device time of the step:
if the first is not well below the second, launch gaps open whenever the host is slow or shared."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from casmtr_amd.pipeline import HotPath, HotPathConfig, make_synthetic_inputs  # noqa: E402


cfg = HotPathConfig.named(sys.argv[1] if len(sys.argv) > 1 else "4c")
model = HotPath(cfg).cuda()
inp = make_synthetic_inputs(cfg, 8, "cuda", seed=1)
for _ in range(3):
    out = model(inp)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model(inp, finalize=False)   # everything enqueued, no read-back of the match counts
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3)
    tot.append((t2 - t0) * 1e3)
enq.sort(); tot.sort()
print(json.dumps({"config": cfg.name, "host_enqueue_ms_median": round(enq[5], 2), "host_enqueue_ms_max": round(enq[-1], 2),
                  "step_ms_median": round(tot[5], 2)}))
