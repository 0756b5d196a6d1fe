"""Times one CascadeQTAttB launch (H=4, K=100, 208x208, B=8) in isolation; CASMTR_QUAD_STOP=n exits after phase n (experiment builds)."""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import ops

B, H, C, hc, wc = 8, 4, 128, 104, 104
h, w = 2 * hc, 2 * wc
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
q, k, v = rn(B, h * w, C), rn(B, h * w, C), rn(B, h * w, C)
# smooth coarse matches (shift by a few cells) like the bench's warped features
ys, xs = torch.meshgrid(torch.arange(hc, device="cuda"), torch.arange(wc, device="cuda"), indexing="ij")
cidx = ((ys + 3).clamp(max=hc - 1) * wc + (xs + 5).clamp(max=wc - 1)).reshape(1, -1).repeat(B, 1)
tp = ops.window_warp_idx(cidx, hc, wc, 5)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: ops.cascade_attn(q, k, v, tp, (h, w), (h, w), H))
print(f"STOP={os.environ.get('CASMTR_QUAD_STOP', '0')}  cascade launch {ms*1e3:.1f} us")
