"""The dual softmax of CasMTR-4c at 832x832 (104^2 tokens, C = 256, 8 pairs) for PMC passes: python tools/ds_only.py [n] [unused] [split|exact]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
gemm = sys.argv[3] if len(sys.argv) > 3 else "split"
B, h, C = 8, 104, 256
g = torch.Generator(device="cuda").manual_seed(0)
f0 = torch.randn((B, h * h, C), generator=g, device="cuda")
f1 = torch.randn((B, h * h, C), generator=g, device="cuda")
torch.cuda.synchronize()
for _ in range(n):
    d = ops.dual_softmax(f0, f1, (h, h), (h, h), 0.1, 0.2, want_conf=False, gemm=gemm)
torch.cuda.synchronize()
print(int(d["n"].item()))
