"""One CascadeMatching window-scoring workload (B = 8, 208 x 208, C = 128, 5 x 5 windows around a smooth coarse match field with 12 %
random matches) for PMC passes and kernel A/B: `python tools/window_only.py [n] [debug_flags] [kernel]` (kernel: pair | dma | quad)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 3:
    os.environ["CASMTR_WINDOW_KERNEL"] = sys.argv[3]
from casmtr_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, C, hc, wc = 8, 128, 104, 104
h, w = 2 * hc, 2 * wc
g = torch.Generator(device="cuda").manual_seed(0)
fq = torch.randn(B, h * w, C, generator=g, device="cuda")
fk = torch.randn(B, h * w, C, generator=g, device="cuda")
ys, xs = torch.meshgrid(torch.arange(hc, device="cuda"), torch.arange(wc, device="cuda"), indexing="ij")
cidx = ((ys + 3).clamp(max=hc - 1) * wc + (xs + 5).clamp(max=wc - 1)).reshape(1, -1).repeat(B, 1)
rnd = torch.rand(cidx.shape, generator=g, device="cuda") < 0.12
cidx = torch.where(rnd, torch.randint(0, hc * wc, cidx.shape, generator=g, device="cuda"), cidx)
wi = ops.WindowIndex(ops.window_warp_idx(cidx, hc, wc, 5), (h, w), (h, w), 1)
for _ in range(n):
    ops.window_match(fq, fk, wi, 1.0, recip=True, want_conf=True)
torch.cuda.synchronize()
if n <= 2:
    sys.exit(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.window_match(fq, fk, wi, 1.0, recip=True, want_conf=True)
e1.record()
torch.cuda.synchronize()
print(f"window_match [{os.environ.get('CASMTR_WINDOW_KERNEL', 'pair')}]: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
