"""L0-shaped launches of the quad-major fine kernel for cache experiments: python tools/fq_exp.py <side> <B> [n]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import ops
side, B = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
H, C, Kp = 8, 256, 16
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
hw = (side, side)
q, k, v = rn(B, side * side, C), rn(B, side * side, C), rn(B, side * side, C)
Lq = Sp = (side // 2) ** 2
prev = torch.stack([torch.argsort(torch.rand(B, Lq, Sp, generator=g, device="cuda"), dim=-1)[..., :Kp] for _ in range(H)], -1).contiguous()
acc = rn(B, Lq, C)
qq, kq, vq, tab = ops.tokens_to_quads(q, *hw), ops.tokens_to_quads(k, *hw), ops.tokens_to_quads(v, *hw), ops.topk_idx_to_tab(prev)
for _ in range(2):
    ops.qta_fine_level_quad(qq, kq, vq, tab, hw, hw, H, 0, w_level=0.3, acc_in=acc, want_message=False, want_topk=False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    ops.qta_fine_level_quad(qq, kq, vq, tab, hw, hw, H, 0, w_level=0.3, acc_in=acc, want_message=False, want_topk=False)
e1.record(); torch.cuda.synchronize()
print(f"side {side} B {B} flags {os.environ.get('CASMTR_FQ_FLAGS','0')} wpx {os.environ.get('CASMTR_FQ_WAVES_PER_XCD','default')}: {e0.elapsed_time(e1)/n*1e3:.1f} us per launch, {B*Lq*H} items")
