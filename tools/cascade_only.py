"""One CascadeQTAttB workload (B=8, 208x208, H=4, K=100, smooth coarse matches) for PMC passes: `python tools/cascade_only.py [n] [debug_flags]`."""
import os
os.environ["CASMTR_DEBUG_HOOKS"] = "1"   # casmtr_debug_set() is ignored without this opt-in
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import _lib, ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
_lib.lib().casmtr_debug_set(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
B, H, C, hc, wc = 8, 4, 128, 104, 104
h, w = 2 * hc, 2 * wc
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
q, k, v = rn(B, h * w, C), rn(B, h * w, C), rn(B, h * w, C)
ys, xs = torch.meshgrid(torch.arange(hc, device="cuda"), torch.arange(wc, device="cuda"), indexing="ij")
cidx = ((ys + 3).clamp(max=hc - 1) * wc + (xs + 5).clamp(max=wc - 1)).reshape(1, -1).repeat(B, 1)
tp = ops.window_warp_idx(cidx, hc, wc, 5)
for _ in range(n):
    ops.cascade_attn(q, k, v, tp, (h, w), (h, w), H, want_idx=False)
torch.cuda.synchronize()
