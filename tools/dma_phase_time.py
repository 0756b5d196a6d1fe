"""Phase-elimination timing of the LDS-DMA kernels (cascade attention, implicit-window matching) at the CasMTR-4c shapes
(B=8, 208x208, C=128, K=100): casmtr_debug_set(1) removes the row transfers, (2) the arithmetic, (3) both.
`--random` draws the coarse matches at random (windows all over the key grid) instead of a smooth shift."""
import os
os.environ["CASMTR_DEBUG_HOOKS"] = "1"   # casmtr_debug_set() is ignored without this opt-in
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import _lib, ops

B, H, C, hc, wc = 8, 4, 128, 104, 104
h, w = 2 * hc, 2 * wc
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
q, k, v = rn(B, h * w, C), rn(B, h * w, C), rn(B, h * w, C)
if "--random" in sys.argv:
    cidx = torch.randint(0, hc * wc, (B, hc * wc), generator=g, device="cuda")
else:
    ys, xs = torch.meshgrid(torch.arange(hc, device="cuda"), torch.arange(wc, device="cuda"), indexing="ij")
    cidx = ((ys + 3).clamp(max=hc - 1) * wc + (xs + 5).clamp(max=wc - 1)).reshape(1, -1).repeat(B, 1)
tp = ops.window_warp_idx(cidx, hc, wc, 5)
wi = ops.WindowIndex(tp, (h, w), (h, w), 1)
full = wi.materialize()


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


tag = f"[{os.environ.get('CASMTR_CASCADE_KERNEL', 'dma')}{' random windows' if '--random' in sys.argv else ''}]"
for flags in (0, 1, 2, 3):
    _lib.lib().casmtr_debug_set(flags)
    a = t(lambda: ops.cascade_attn(q, k, v, tp, (h, w), (h, w), H, want_idx=False))
    m = t(lambda: ops.window_match(q, k, wi, 1.0, want_conf=True))
    print(f"{tag} debug flags {flags}: cascade_attn {a:7.1f} us   window_match(implicit) {m:7.1f} us")
_lib.lib().casmtr_debug_set(0)
print(f"{tag} cascade_attn + up_idx write {t(lambda: ops.cascade_attn(q, k, v, tp, (h, w), (h, w), H, want_idx=True)):7.1f} us   "
      f"window_match(explicit idx) {t(lambda: ops.window_match(q, k, full, 1.0, want_conf=True, hw=(h, w))):7.1f} us")
