"""where the indoor model's 1/4 stage spends its time (640x480, batch 8)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from casmtr_amd.model import CasMTRIndoor4c
from casmtr_amd.model.casmtr4c import _grid, _tokens
from casmtr_amd import ops

torch.manual_seed(0)
m = CasMTRIndoor4c().eval().cuda()
B = 8
im0, im1 = torch.rand(B, 3, 480, 640, device="cuda"), torch.rand(B, 3, 480, 640, device="cuda")


def T(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r


with torch.no_grad():
    data = {"image0": im0, "image1": im1}
    x, f8, f4, ff = m.features(data)
    t8 = m.coarse_stage(f8, data)
    ms, (l4, lf) = T(lambda: m.ladder(x, f4, ff)); print(f"ladder {ms:.2f} ms")
    ms, f4_0 = T(lambda: m.up_block1(l4[:B], _grid(t8[0], *data["hw0_8c"]))); print(f"up_block (one image set) {ms:.2f} ms")
    f4_1 = m.up_block1(l4[B:], _grid(t8[1], *data["hw1_8c"]))
    tr = m.loftr_coarse_4c
    st8 = data["stage_8c"]
    H, W = f4_0.shape[2:]
    tp01 = ops.window_warp_idx(st8["next_idx_c01"].contiguous(), H // 2, W // 2, 5)
    ms, rp = T(lambda: tr.relative_pe(st8["next_idx_c01"], tp01, data["hw0_8c"], data["hw1_8c"], H)); print(f"relative_pe (one direction) {ms:.2f} ms")
    a, b = _tokens(m.pos_encoding_4c(f4_0)).contiguous(), _tokens(m.pos_encoding_4c(f4_1)).contiguous()
    ms, _ = T(lambda: tr.layers[0](a, H, W)); print(f"POLA block (one image set) {ms:.2f} ms")
    ms, _ = T(lambda: tr.layers[1](a, b, H, W, H, W, tp01, rel_pos=rp)); print(f"cascade cross block with rel_pos (one direction) {ms:.2f} ms")
    ms, _ = T(lambda: tr.layers[1](a, b, H, W, H, W, tp01, rel_pos=None)); print(f"cascade cross block without rel_pos {ms:.2f} ms")
