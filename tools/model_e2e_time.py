#!/usr/bin/env python3
"""Whole-model timing of casmtr_amd.model.CasMTR4c: random-init weights, synthetic 832x832 pairs, per-stage
HIP-event breakdown (backbone / 1/8 stage / 1/4 stage / fine stage).  A SECOND metric next to bench.py's hot-path line: it
includes the torch-op glue (backbone convolutions, MLPs, window self-attention) that bench.py leaves out.

    python tools/model_e2e_time.py [--batch 8] [--size 832] [--steps 10] [--cascade-thr 0.2]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from casmtr_amd.model import CasMTR4c, outdoor_4c_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=832)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--coarse-thr", type=float, default=None)
    ap.add_argument("--cascade-thr", type=float, default=None)
    a = ap.parse_args()
    cfg = outdoor_4c_config()
    if a.coarse_thr is not None:
        cfg["match_coarse"]["thr"] = a.coarse_thr
    if a.cascade_thr is not None:
        cfg["match_cascade"].update(test_thr=a.cascade_thr, pre_thr=[0.0])
    torch.manual_seed(0)
    m = CasMTR4c(cfg).eval().cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    mk = lambda: torch.rand((a.batch, 3, a.size, a.size), device="cuda", generator=g)
    sets = [(mk(), mk()) for _ in range(2)]
    names = ["backbone", "stage_8c", "stage_4c", "fine"]
    acc = dict.fromkeys(names, 0.0)
    nm = 0

    def step(i, timed):
        nonlocal nm
        im0, im1 = sets[i % 2]
        data = {"image0": im0, "image1": im1}
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        with torch.no_grad():
            ev[0].record()
            (f8_0, f8_1), (f4_0, f4_1), (ff0, ff1) = m.features(data)
            ev[1].record()
            t8 = m.coarse_stage(f8_0, f8_1, data)
            ev[2].record()
            t4 = m.cascade_stage(f4_0, f4_1, *t8, data)
            ev[3].record()
            m.fine_stage(ff0, ff1, *t4, data)
            ev[4].record()
        torch.cuda.synchronize()
        if timed:
            for k, n in enumerate(names):
                acc[n] += ev[k].elapsed_time(ev[k + 1])
            nm += int(data["mkpts0_f"].shape[0])

    for i in range(a.warmup):
        step(i, False)
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for i in range(a.steps):
        step(i, True)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / a.steps
    print(json.dumps({"metric": "whole-model image pairs/sec (CasMTR-4c, torch glue + HIP hot path)", "value": round(a.batch / ms * 1e3, 2),
                      "ms_per_step": round(ms, 2), "batch": a.batch, "size": a.size, "steps": a.steps,
                      "stage_ms": {k: round(v / a.steps, 2) for k, v in acc.items()}, "matches_per_pair": round(nm / a.steps / a.batch, 1),
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "data": "synthetic", "weights": "random-init"}))


if __name__ == "__main__":
    main()
