"""CascadeQTAttB launch time (B=8, 208x208, H=4, K=100) for smooth coarse matches (what real image pairs produce: neighbouring
queries look at neighbouring windows) against uniformly random ones (the benchmark's random features), and for the random case
with the queries' work order sorted by window position -- how much of the kernel's time is the cache-unfriendly order."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import ops

B, H, C, hc, wc = 8, 4, 128, 104, 104
h, w = 2 * hc, 2 * wc
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
q, k, v = rn(B, h * w, C), rn(B, h * w, C), rn(B, h * w, C)
ys, xs = torch.meshgrid(torch.arange(hc, device="cuda"), torch.arange(wc, device="cuda"), indexing="ij")
smooth = ((ys + 3).clamp(max=hc - 1) * wc + (xs + 5).clamp(max=wc - 1)).reshape(1, -1).repeat(B, 1)
rand = torch.randint(0, hc * wc, (B, hc * wc), generator=g, device="cuda")


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


for name, cidx in (("smooth", smooth), ("random", rand)):
    tp = ops.window_warp_idx(cidx, hc, wc, 5)
    print(f"{name}: cascade_attn {t(lambda: ops.cascade_attn(q, k, v, tp, (h, w), (h, w), H, want_idx=False)):.1f} us")
# random matches, but the QUERY tokens re-ordered so that consecutive quads look at neighbouring windows: an upper bound for what a
# sorted work list could give (the permuted q stands in for an indirection inside the kernel)
order = torch.argsort(rand, dim=1)                                   # coarse query cells sorted by their target cell
cs = torch.gather(rand, 1, order)
tp = ops.window_warp_idx(cs, hc, wc, 5)
print(f"random, sorted by target: cascade_attn {t(lambda: ops.cascade_attn(q, k, v, tp, (h, w), (h, w), H, want_idx=False)):.1f} us")
