import os, sys, torch
sys.path.insert(0, "/root/repo")
from casmtr_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
B, H, C = 8, 8, 256
for level, (side, Kp, topk) in enumerate(((104, 16, 0), (52, 32, 16))):
    hw = (side, side)
    q, k, v = rn(B, side * side, C), rn(B, side * side, C), rn(B, side * side, C)
    Lq = (side // 2) ** 2
    prev = torch.stack([torch.argsort(torch.rand(B, Lq, Lq, generator=g, device="cuda"), dim=-1)[..., :Kp] for _ in range(H)], -1).contiguous()
    acc = rn(B, Lq, C)
    qq, kq, vq, tab = ops.tokens_to_quads(q, *hw), ops.tokens_to_quads(k, *hw), ops.tokens_to_quads(v, *hw), ops.topk_idx_to_tab(prev)
    run = lambda: ops.qta_fine_level_quad(qq, kq, vq, tab, hw, hw, H, topk, w_level=0.3, acc_in=acc, want_message=False, want_topk=False)
    for rep in range(2):
        for fl in ("0", "32", "8", "40"):
            os.environ["CASMTR_FQ_FLAGS"] = fl
            for _ in range(3): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            print(f"level {level} flags {fl:>2s} (32 = no stores, 8 = one K/V slice): {e0.elapsed_time(e1)/20*1e3:.1f} us", flush=True)
