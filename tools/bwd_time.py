"""Times the primitive ops and their backward kernels (a12 / f.2) at the CasMTR-4c 832x832 shapes, one pair:
python tools/bwd_time.py [B]  -> us per launch and the rate of the bytes each launch must move."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")


def t(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N1, N2, K, H in (("QTAttB level 1 (52x52, K=128, H=8)", 676, 2704, 128, 8), ("QTAttB level 0 (104x104, K=64, H=8)", 2704, 10816, 64, 8),
                           ("CascadeQTAttB (208x208, K=100, H=4)", 10816, 43264, 100, 4)):
    D = 32
    q, key, val = rn(B, N1, 4, H, D), rn(B, N2, H, D), rn(B, N2, H, D)
    idx = torch.randint(0, N2, (B, N1, K, H), generator=g, device="cuda")
    gs = rn(B, N1, 4, K, H)
    sc = torch.softmax(gs, dim=3).reshape(B, N1 * 4, K, H).contiguous()
    idx5 = idx[:, :, None].expand(-1, -1, 4, -1, -1).reshape(B, N1 * 4, K, H).contiguous()
    out = torch.zeros((B, N1 * 4, H, D), device="cuda")
    go = rn(B, N1 * 4, H, D)
    gsc, gv = torch.zeros_like(sc), torch.zeros_like(val)
    tf = t(lambda: ops.qta_score_fwd(q, key, idx))
    tb = t(lambda: ops.qta_score_bwd(gs, q, key, idx))
    vf = t(lambda: ops.qta_value_agg_fwd(sc, val, idx5, out))
    vb = t(lambda: (gsc.zero_(), gv.zero_(), ops.qta_value_agg_bwd(go, sc, val, idx5, gsc, gv)))
    gath = B * N1 * K * H * D * 4 / 1e9          # gathered key / value bytes per launch
    print(f"{name}, B={B}: score fwd {tf:.0f} us, bwd {tb:.0f} us | value_agg fwd {vf:.0f} us, bwd {vb:.0f} us (incl. 2 zero fills) | "
          f"gathered rows {gath:.2f} GB per launch -> fwd {gath / tf * 1e3:.1f} / bwd {gath / tb * 1e3:.1f} TB/s")
N, K, C = 43264, 100, 128
q, key = rn(B, N, C), rn(B, N, C)
idx = torch.randint(0, N, (B, N, K), generator=g, device="cuda")
gw = rn(B, N, K)
wf = t(lambda: ops.window_score_fwd(q, key, idx))
wb = t(lambda: ops.window_score_bwd(gw, q, key, idx))
gath = B * N * K * C * 4 / 1e9
print(f"score_cuda window scoring (208x208, K=100, C=128), B={B}: fwd {wf:.0f} us, bwd {wb:.0f} us | gathered rows {gath:.2f} GB -> "
      f"fwd {gath / wf * 1e3:.1f} / bwd {gath / wb * 1e3:.1f} TB/s")
