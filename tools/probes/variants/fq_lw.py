"""Loader-wave kernel of the finest QTAttB level (csrc/fine_lw.hip, CASMTR_FQ_LW=1) against fine_quad_kernel<1,false,true>: bit-equality of
the results and us per launch as a function of the persistent grid (CASMTR_LW_BLOCKS workgroups of 1 loader + 2 consumer waves)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import ops

B, H, side, Kp = int(os.environ.get("FQ_B", 8)), int(os.environ.get("FQ_H", 8)), int(os.environ.get("FQ_SIDE", 104)), 16
C = 32 * H
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
hw = (side, side)
q, k, v = rn(B, side * side, C), rn(B, side * side, C), rn(B, side * side, C)
Lq = Sp = (side // 2) ** 2
prev = torch.stack([torch.argsort(torch.rand(B, Lq, Sp, generator=g, device="cuda"), dim=-1)[..., :Kp] for _ in range(H)], -1).contiguous()
acc = rn(B, Lq, C)
qq, kq, vq, tab = ops.tokens_to_quads(q, *hw), ops.tokens_to_quads(k, *hw), ops.tokens_to_quads(v, *hw), ops.topk_idx_to_tab(prev)
run = lambda: ops.qta_fine_level_quad(qq, kq, vq, tab, hw, hw, H, 0, w_level=0.3, acc_in=acc, want_message=True, want_topk=False)
os.environ["CASMTR_FQ_VARIANT"] = ""
ref = run()
torch.cuda.synchronize()


def timeit(n=20):
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"fine_quad_kernel: {timeit():.1f} us per launch", flush=True)
def check(label):
    poison = [torch.full_like(ref[kk], float("nan")) for kk in ("acc", "message") if ref.get(kk) is not None]   # the allocator hands these blocks to run()
    torch.cuda.synchronize()
    del poison
    out = run()
    torch.cuda.synchronize()
    same = all(torch.equal(out[kk], ref[kk]) for kk in ("acc", "message") if ref.get(kk) is not None and out.get(kk) is not None)
    md = max(float((out[kk] - ref[kk]).abs().max()) for kk in ("acc", "message") if ref.get(kk) is not None)
    print(f"{label}: bit-equal {same} (max abs diff {md:.2e}); {timeit():.1f} us per launch", flush=True)


os.environ["CASMTR_FQ_VARIANT"] = "vs"
for blocks in (None, 640, 512, 384):
    if blocks is None:
        os.environ.pop("CASMTR_VS_BLOCKS", None)
    else:
        os.environ["CASMTR_VS_BLOCKS"] = str(blocks)
    check(f"fine_vs_kernel, {blocks or 'resident'} workgroups")
if os.environ.get("FQ_LW_SKIP"):
    sys.exit(0)
os.environ["CASMTR_FQ_VARIANT"] = "lw"
for nc, blocks in ((2, None), (2, 512), (1, None), (1, 1024), (3, None)):
    os.environ["CASMTR_LW_NC"] = str(nc)
    if blocks is None:
        os.environ.pop("CASMTR_LW_BLOCKS", None)
    else:
        os.environ["CASMTR_LW_BLOCKS"] = str(blocks)
    poison = [torch.full_like(ref[kk], float("nan")) for kk in ("acc", "message") if ref.get(kk) is not None]   # the allocator hands these blocks to run()
    torch.cuda.synchronize()
    del poison
    out = run()
    torch.cuda.synchronize()
    same = all(torch.equal(out[kk], ref[kk]) for kk in ("acc", "message") if ref.get(kk) is not None and out.get(kk) is not None)
    md = max(float((out[kk] - ref[kk]).abs().max()) for kk in ("acc", "message") if ref.get(kk) is not None)
    print(f"fine_lw_kernel, {nc} consumers per loader, {blocks or 'resident'} workgroups: bit-equal {same} (max abs diff {md:.2e}); {timeit():.1f} us per launch", flush=True)
