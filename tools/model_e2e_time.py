#!/usr/bin/env python3
"""Whole-model timing of casmtr_amd.model.CasMTR4c; see casmtr_amd/model/timing.py.

    python tools/model_e2e_time.py [--batch 8] [--size 832] [--steps 10] [--cascade-thr 0.2]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from casmtr_amd.model.timing import time_whole_model  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=832)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--coarse-thr", type=float, default=None)
    ap.add_argument("--cascade-thr", type=float, default=None)
    ap.add_argument("--model", choices=["4c", "2c", "indoor"], default="4c")
    ap.add_argument("--conv-dtype", choices=["fp32", "fp16", "bf16"], default="fp32")
    a = ap.parse_args()
    print(json.dumps(time_whole_model(a.batch, a.size, a.steps, a.warmup, a.coarse_thr, a.cascade_thr, model=a.model,
                                      conv_dtype={"fp32": None, "fp16": torch.float16, "bf16": torch.bfloat16}[a.conv_dtype])))
