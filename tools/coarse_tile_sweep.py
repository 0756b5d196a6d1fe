"""Sweep of the register-tile coarsest-level kernel (csrc/coarse_tile.hip) at 26x26, H = 8, top-32, B = 8: ring slots x
phase-elimination flags (256 no selection, 512 no A.V arithmetic, 1024 no DMA, 2048 no barriers).  us per call.  (The 8-wave and 6-slot
instances of the first sweeps are no longer built: DESIGN.md section 12 has their numbers.)"""
import os
os.environ["CASMTR_DEBUG_HOOKS"] = "1"
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import _lib, ops

B, H, C, L = 8, 8, 256, 676
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn(B, L, C, generator=g, device="cuda") for _ in range(3))


def t(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


flagsets = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,256,512,768,1024,1792".split(","))]
for nw in (4,):
    for ns in (3, 4):
        os.environ["CASMTR_CT_SLOTS"] = str(ns)
        row = []
        for flags in flagsets:
            _lib.lib().casmtr_debug_set(flags)
            row.append(f"{flags}: {t(lambda: ops.qta_coarse_level(q, k, v, H, 32, w_level=0.3, want_message=False, want_tab=True, want_topk=False)):6.1f}")
        print(f"waves {nw} slots {ns} | " + " | ".join(row))
_lib.lib().casmtr_debug_set(0)
