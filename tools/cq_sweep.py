"""Isolated timing of the quad-major cascade attention kernel (208x208, H = 4, 5x5 windows on smooth coarse matches, B = 8) as a
function of the persistent grid size: python tools/cq_sweep.py -> us per launch per CASMTR_CQ_WAVES_PER_XCD."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import ops

B, H, C, hc, wc = 8, 4, 128, 104, 104
h, w = 2 * hc, 2 * wc
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
q, k, v = (ops.tokens_to_quads(rn(B, h * w, C), h, w) for _ in range(3))
ys, xs = torch.meshgrid(torch.arange(hc, device="cuda"), torch.arange(wc, device="cuda"), indexing="ij")
cidx = ((ys + 3).clamp(max=hc - 1) * wc + (xs + 5).clamp(max=wc - 1)).reshape(1, -1).repeat(B, 1)
tp = ops.window_warp_idx(cidx, hc, wc, 5)
for wpx in (None, 512, 448, 384, 320, 256, 192, 128):
    if wpx is None:
        os.environ.pop("CASMTR_CQ_WAVES_PER_XCD", None)
    else:
        os.environ["CASMTR_CQ_WAVES_PER_XCD"] = str(wpx)
    for _ in range(3):
        ops.cascade_attn_quad(q, k, v, tp, (h, w), (h, w), H)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.cascade_attn_quad(q, k, v, tp, (h, w), (h, w), H)
    e1.record()
    torch.cuda.synchronize()
    print(f"cascade_quad: waves per XCD {wpx if wpx else 'default'}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch", flush=True)
