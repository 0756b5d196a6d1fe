"""Runs only the QTAttB fine levels (K=128 then K=64 launches) at the BASELINE shape, for PMC passes on quad_attn_kernel."""
import sys
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import ops

B, H, C = 8, 8, 256
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
q2, k2, v2 = rn(B, 676, C), rn(B, 676, C), rn(B, 676, C)
q1, k1, v1 = rn(B, 2704, C), rn(B, 2704, C), rn(B, 2704, C)
q0, k0, v0 = rn(B, 10816, C), rn(B, 10816, C), rn(B, 10816, C)
l0 = ops.qta_coarse_level(q2, k2, v2, H, 32, w_level=0.3, want_message=False)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(reps):
    l1 = ops.qta_fine_level(q1, k1, v1, l0["topk_idx"], (52, 52), (52, 52), H, 16, w_level=0.3, acc_in=l0["acc"], want_message=False)
    l2 = ops.qta_fine_level(q0, k0, v0, l1["topk_idx"], (104, 104), (104, 104), H, 0, w_level=0.4, acc_in=l1["acc"], want_message=False)
torch.cuda.synchronize()
print("ok", float(l2["acc"].abs().mean()))
