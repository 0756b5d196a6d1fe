"""L0-shaped launches of the quad-major fine kernel for cache experiments: python tools/fq_exp.py <side> <B> [n]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import ops
side, B = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
H = int(sys.argv[4]) if len(sys.argv) > 4 else 8
Kp = int(sys.argv[5]) if len(sys.argv) > 5 else 16
C = 32 * H
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
hw = (side, side)
q, k, v = rn(B, side * side, C), rn(B, side * side, C), rn(B, side * side, C)
Lq = Sp = (side // 2) ** 2
windows = os.environ.get("FQ_WINDOWS", "0") == "1"
if windows:   # cascade-like parents: the 5x5 window (shifted inside the grid) around the quad's own cell moved by (3, 5); same for all heads
    hq = side // 2
    qy, qx = torch.meshgrid(torch.arange(hq, device="cuda"), torch.arange(hq, device="cuda"), indexing="ij")
    oy = (qy + 3 - 2).clamp(0, hq - 5).reshape(-1)
    ox = (qx + 5 - 2).clamp(0, hq - 5).reshape(-1)
    e = torch.arange(25, device="cuda")
    cells = (oy[:, None] + e[None, :] // 5) * hq + ox[:, None] + e[None, :] % 5           # [Lq, 25]
    tab = cells[None, None].expand(B, H, Lq, 25).to(torch.int32).contiguous()
    assert Kp == 25
else:
    prev = torch.stack([torch.argsort(torch.rand(B, Lq, Sp, generator=g, device="cuda"), dim=-1)[..., :Kp] for _ in range(H)], -1).contiguous()
    tab = ops.topk_idx_to_tab(prev)
acc = rn(B, Lq, C)
qq, kq, vq = ops.tokens_to_quads(q, *hw), ops.tokens_to_quads(k, *hw), ops.tokens_to_quads(v, *hw)
for _ in range(2):
    ops.qta_fine_level_quad(qq, kq, vq, tab, hw, hw, H, 0, w_level=0.3, acc_in=acc, want_message=False, want_topk=False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    ops.qta_fine_level_quad(qq, kq, vq, tab, hw, hw, H, 0, w_level=0.3, acc_in=acc, want_message=False, want_topk=False)
e1.record(); torch.cuda.synchronize()
print(f"H {H} Kp {Kp} side {side} B {B} flags {os.environ.get('CASMTR_FQ_FLAGS','0')} wpx {os.environ.get('CASMTR_FQ_WAVES_PER_XCD','default')}: {e0.elapsed_time(e1)/n*1e3:.1f} us per launch, {B*Lq*H} items")
