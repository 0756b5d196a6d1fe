"""Isolated timing of the quad-major fine-level kernel at the CasMTR-4c shapes (B = 8) as a function of the persistent grid size:
python tools/fq_sweep.py   -> us per launch for level 0 (104x104, lists of 64, no top-k) and level 1 (52x52, lists of 128, top-16)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import ops

B, H, C = 8, 8, 256
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
for level, (side, Kp, topk) in enumerate(((104, 16, 0), (52, 32, 16))):
    hw = (side, side)
    q, k, v = rn(B, side * side, C), rn(B, side * side, C), rn(B, side * side, C)
    Lq = Sp = (side // 2) ** 2
    prev = torch.stack([torch.argsort(torch.rand(B, Lq, Sp, generator=g, device="cuda"), dim=-1)[..., :Kp] for _ in range(H)], -1).contiguous()
    acc = rn(B, Lq, C)
    qq, kq, vq, tab = ops.tokens_to_quads(q, *hw), ops.tokens_to_quads(k, *hw), ops.tokens_to_quads(v, *hw), ops.topk_idx_to_tab(prev)
    # keep other (pair-sized) tensors cycling through the caches between launches, as the real step does
    for wpx in (512, None, 512, 480, 448, 416, 384, 352, 320, 256):
        if wpx is None:
            os.environ.pop("CASMTR_FQ_WAVES_PER_XCD", None)
        else:
            os.environ["CASMTR_FQ_WAVES_PER_XCD"] = str(wpx)
        for _ in range(3):
            ops.qta_fine_level_quad(qq, kq, vq, tab, hw, hw, H, topk, w_level=0.3, acc_in=acc, want_message=False, want_topk=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.qta_fine_level_quad(qq, kq, vq, tab, hw, hw, H, topk, w_level=0.3, acc_in=acc, want_message=False, want_topk=False)
        e1.record()
        torch.cuda.synchronize()
        print(f"level {level}: waves per XCD {wpx if wpx else 'default'}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch", flush=True)
