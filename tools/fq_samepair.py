"""Does the finest QTAttB level lose its extra waves to the (pair, head) slice transitions in the L2?  Same sweep as tools/fq_sweep.py, once
as it is and once with CASMTR_FQ_FLAGS=8 (every pair gathers from pair 0's K / V slices: one 2.8 MB slice per XCD for the whole launch;
results are garbage, timing only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import ops

B, H, C, side, Kp = 8, 8, 256, 104, 16
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
hw = (side, side)
q, k, v = rn(B, side * side, C), rn(B, side * side, C), rn(B, side * side, C)
Lq = Sp = (side // 2) ** 2
prev = torch.stack([torch.argsort(torch.rand(B, Lq, Sp, generator=g, device="cuda"), dim=-1)[..., :Kp] for _ in range(H)], -1).contiguous()
acc = rn(B, Lq, C)
qq, kq, vq, tab = ops.tokens_to_quads(q, *hw), ops.tokens_to_quads(k, *hw), ops.tokens_to_quads(v, *hw), ops.topk_idx_to_tab(prev)
LABEL = {"8": "one slice for all pairs", "0": "as shipped"}
for flags in ("0", "8"):
    os.environ["CASMTR_FQ_FLAGS"] = flags
    for wpx in (256, 320, 384):
        os.environ["CASMTR_FQ_WAVES_PER_XCD"] = str(wpx)
        for _ in range(3):
            ops.qta_fine_level_quad(qq, kq, vq, tab, hw, hw, H, 0, w_level=0.3, acc_in=acc, want_message=False, want_topk=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.qta_fine_level_quad(qq, kq, vq, tab, hw, hw, H, 0, w_level=0.3, acc_in=acc, want_message=False, want_topk=False)
        e1.record()
        torch.cuda.synchronize()
        print(f"flags {flags} ({LABEL[flags]}): waves per XCD {wpx}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch", flush=True)
