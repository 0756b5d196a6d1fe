"""The coarsest QTAttB level alone (26x26, H = 8, top-32, B = 8) for PMC passes: `python tools/coarse_only.py [n] [debug_flags] [kernel]`
kernel: tile (default, csrc/coarse_tile.hip) | three | fused."""
import os
os.environ["CASMTR_DEBUG_HOOKS"] = "1"   # casmtr_debug_set() is ignored without this opt-in
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import _lib, ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
_lib.lib().casmtr_debug_set(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
os.environ["CASMTR_COARSE_KERNEL"] = sys.argv[3] if len(sys.argv) > 3 else "tile"
B, H, C, L = 8, 8, 256, 676
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn(B, L, C, generator=g, device="cuda") for _ in range(3))
torch.cuda.synchronize()
for _ in range(n):
    ops.qta_coarse_level(q, k, v, H, 32, w_level=0.3, want_message=False, want_tab=True, want_topk=False)
torch.cuda.synchronize()
