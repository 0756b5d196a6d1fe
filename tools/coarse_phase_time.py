"""Times the coarsest QTAttB level (26x26, H=8, top-32, B=8) for the register-tile kernel (default), the three-kernel path and the
LDS-tile kernel; casmtr_debug_set bits: fused kernel 1 = stop after the logits phase, 2 = skip the row phase, 4 = stop before A.V;
tile kernel 256 = skip the selection, 512 = skip the A.V arithmetic."""
import os
os.environ["CASMTR_DEBUG_HOOKS"] = "1"   # casmtr_debug_set() is ignored without this opt-in
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import _lib, ops

B, H, C, L = 8, 8, 256, 676
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn(B, L, C, generator=g, device="cuda") for _ in range(3))


def t(fn, n=20):
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


for kern, flagset in (("tile", (0, 256, 512, 768)), ("three", (0,))):
    os.environ["CASMTR_COARSE_KERNEL"] = kern
    for flags in flagset:
        _lib.lib().casmtr_debug_set(flags)
        print(f"[{kern}] debug flags {flags}: {t(lambda: ops.qta_coarse_level(q, k, v, H, 32, w_level=0.3, want_message=False, want_tab=True, want_topk=False)):7.1f} us")
_lib.lib().casmtr_debug_set(0)
