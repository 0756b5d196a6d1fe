"""Ablation timing of the finest QTAttB level on the VGPR-staged variant (csrc/fine_vs.hip, CASMTR_FQ_VARIANT=vs): us per launch with one
stage of the item loop removed at a time (CASMTR_VS_ABLATE; the results are wrong, only the time is meaningful), at 8 and 12 waves per
CU.  What a stage costs = the baseline minus the run without it."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import ops

B, H, side, Kp = 8, 8, 104, 16
C = 32 * H
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
hw = (side, side)
q, k, v = rn(B, side * side, C), rn(B, side * side, C), rn(B, side * side, C)
Lq = (side // 2) ** 2
prev = torch.stack([torch.argsort(torch.rand(B, Lq, Lq, generator=g, device="cuda"), dim=-1)[..., :Kp] for _ in range(H)], -1).contiguous()
acc = rn(B, Lq, C)
qq, kq, vq, tab = ops.tokens_to_quads(q, *hw), ops.tokens_to_quads(k, *hw), ops.tokens_to_quads(v, *hw), ops.topk_idx_to_tab(prev)
run = lambda: ops.qta_fine_level_quad(qq, kq, vq, tab, hw, hw, H, 0, w_level=0.3, acc_in=acc, want_message=True, want_topk=False)


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


os.environ["CASMTR_FQ_VARIANT"] = ""
print(f"fine_quad_kernel<1,false,true>: {timeit():.1f} us per launch", flush=True)
os.environ["CASMTR_FQ_VARIANT"] = "vs"
NAMES = {0: "everything", 2: "no softmax", 4: "no V pass (1 of 32 reads and MFMAs)", 8: "no K pass (1 of 16 reads, 4 of 32 MFMAs)", 16: "no row loads",
         32: "no row writes to LDS (2 of 16)", 64: "no output stores", 126: "none of these (loop, staging, waits)",
         46: "memory only: row loads + stores (no softmax, passes, row writes)", 80: "compute only: no row loads, no stores",
         128: "V rows read from the K slice (same accesses, half the L2 working set)"}
only = [int(x) for x in os.environ.get("FQ_ABL_LIST", "").split(",") if x]   # e.g. FQ_ABL_LIST=0,128 under rocprofv3 --pmc
for blocks in ((512,) if only else (512, 768)):
    os.environ["CASMTR_VS_BLOCKS"] = str(blocks)
    for abl, name in NAMES.items():
        if only and abl not in only:
            continue
        os.environ["CASMTR_VS_ABLATE"] = str(abl)
        print(f"fine_vs_kernel, {blocks // 64} waves per CU, {name}: {timeit():.1f} us per launch", flush=True)
