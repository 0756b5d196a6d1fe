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
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.cascade_attn(q, k, v, tp, (h, w), (h, w), H, want_idx=False)
e1.record()
torch.cuda.synchronize()
print(f"cascade_attn (current kernel), smooth windows: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
qm = ops.nchw_to_quads_multi([x.view(B, h, w, C).permute(0, 3, 1, 2).contiguous() for x in (q, k, v)])
for wpx in (None, 512, 384, 256, 192, 128):
    if wpx: os.environ["CASMTR_CQ_WAVES_PER_XCD"] = str(wpx)
    for _ in range(3):
        ops.cascade_attn_quad(qm[0], qm[1], qm[2], tp, (h, w), (h, w), H)
    e0.record()
    for _ in range(20):
        ops.cascade_attn_quad(qm[0], qm[1], qm[2], tp, (h, w), (h, w), H)
    e1.record()
    torch.cuda.synchronize()
    print(f"cascade_attn_quad (pair kernel), smooth windows, waves per XCD {wpx}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
os.environ.pop("CASMTR_CQ_WAVES_PER_XCD", None)
m_old, _ = ops.cascade_attn(q, k, v, tp, (h, w), (h, w), H, want_idx=False)
m_new = ops.cascade_attn_quad(qm[0], qm[1], qm[2], tp, (h, w), (h, w), H)
print("max abs diff old vs new:", float((m_old - m_new).abs().max()))
