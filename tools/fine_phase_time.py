"""Times the two fine-level launches of one QTAttB call at the CasMTR-4c shapes (B=8, H=8: 52x52 with K=128 / top-16 and 104x104
with K=64, no top-k) on random previous-level indices, for both kernels and with the phase-elimination switches."""
import os
os.environ["CASMTR_DEBUG_HOOKS"] = "1"   # casmtr_debug_set() is ignored without this opt-in
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import _lib, ops

B, H, C = 8, 8, 256
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")


def t(fn, n=10):
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


lv = {}
for name, side, Kp, topk in (("L1 52x52 K=128 top16", 52, 32, 16), ("L0 104x104 K=64", 104, 16, 0)):
    q, k, v = rn(B, side * side, C), rn(B, side * side, C), rn(B, side * side, C)
    Lq, Sp = (side // 2) ** 2, (side // 2) ** 2
    prev = torch.stack([torch.argsort(torch.rand(B, Lq, Sp, generator=g, device="cuda"), dim=-1)[..., :Kp] for _ in range(H)], -1).contiguous()
    acc = rn(B, Lq, C)
    lv[name] = (q, k, v, prev, side, topk, acc)
for kern in ("dma", "quad"):
    os.environ["CASMTR_FINE_KERNEL"] = kern
    for flags in ((0, 1, 2, 3) if kern == "dma" else (0,)):  # the phase switches exist in the dma kernel only
        _lib.lib().casmtr_debug_set(flags)
        out = []
        for name, (q, k, v, prev, side, topk, acc) in lv.items():
            us = t(lambda: ops.qta_fine_level(q, k, v, prev, (side, side), (side, side), H, topk, w_level=0.3, acc_in=acc, want_message=False))
            out.append(f"{name}: {us:7.1f} us")
        print(f"[{kern}] debug flags {flags}:  " + "   ".join(out))
_lib.lib().casmtr_debug_set(0)
