"""One fine-level workload of QTAttB at the CasMTR-4c shapes for PMC passes: `python tools/fine_only.py [n] [debug_flags] [level] [path]`
level 0: 104x104, lists of 64, no top-k; level 1: 52x52, lists of 128, top-16.
path qm (default): the round-3 quad-major kernel fine_quad_kernel<1,false> / <2,true>; tok: the round-2 token-major kernels
(fine_level_dma_kernel / quad_attn_kernel<8,128,0>)."""
import os
os.environ["CASMTR_DEBUG_HOOKS"] = "1"   # casmtr_debug_set() is ignored without this opt-in
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import _lib, ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
_lib.lib().casmtr_debug_set(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
level = int(sys.argv[3]) if len(sys.argv) > 3 else 0
path = sys.argv[4] if len(sys.argv) > 4 else "qm"
B, H, C = 8, 8, 256
side, Kp, topk = ((104, 16, 0), (52, 32, 16))[level]
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
q, k, v = rn(B, side * side, C), rn(B, side * side, C), rn(B, side * side, C)
Lq = Sp = (side // 2) ** 2
prev = torch.stack([torch.argsort(torch.rand(B, Lq, Sp, generator=g, device="cuda"), dim=-1)[..., :Kp] for _ in range(H)], -1).contiguous()
acc = rn(B, Lq, C)
hw = (side, side)
if path == "qm":
    qq, kq, vq, tab = ops.tokens_to_quads(q, *hw), ops.tokens_to_quads(k, *hw), ops.tokens_to_quads(v, *hw), ops.topk_idx_to_tab(prev)
    torch.cuda.synchronize()
    for _ in range(n):
        ops.qta_fine_level_quad(qq, kq, vq, tab, hw, hw, H, topk, w_level=0.3, acc_in=acc, want_message=False, want_topk=False)
else:
    os.environ["CASMTR_FINE_KERNEL"] = "dma" if level == 0 else "quad"
    for _ in range(n):
        ops.qta_fine_level(q, k, v, prev, hw, hw, H, topk, w_level=0.3, acc_in=acc, want_message=False)
torch.cuda.synchronize()
