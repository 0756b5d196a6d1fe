"""cascade_quad_kernel on controlled window fields (208 x 208, H = 4, B = 8, 5 x 5 windows): fully coherent (a shift), fully random, and
the bench's mix (a shift with a fraction of random cells).  Prints us per launch for the static item schedule (CASMTR_CQ_DYNAMIC=0)
against dynamic claiming, both item orders, and checks that the two schedules give bit-identical messages.
    python tools/cq_fields.py                      # timing table
    python tools/cq_fields.py one <field> <dyn>    # 3 launches of one configuration (for rocprofv3 --pmc FETCH_SIZE passes)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import ops

B, H, C, hc, wc = 8, 4, 128, 104, 104
h, w = 2 * hc, 2 * wc
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
q, k, v = (ops.tokens_to_quads(rn(B, h * w, C), h, w) for _ in range(3))
ys, xs = torch.meshgrid(torch.arange(hc, device="cuda"), torch.arange(wc, device="cuda"), indexing="ij")
smooth = ((ys + 3).clamp(max=hc - 1) * wc + (xs + 5).clamp(max=wc - 1)).reshape(1, -1).repeat(B, 1)
rand = torch.randint(0, hc * wc, (B, hc * wc), generator=g, device="cuda")


def field(name):
    if name == "smooth":
        return smooth
    if name == "random":
        return rand
    frac = float(name[3:]) / 100.0                       # "mix12": 12 % of the cells random, the rest the shift
    pick = torch.rand((B, hc * wc), generator=g, device="cuda") < frac
    return torch.where(pick, rand, smooth)


def run(tp):
    return ops.cascade_attn_quad(q, k, v, tp, (h, w), (h, w), H)


def timeit(tp, n=20):
    for _ in range(3):
        run(tp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run(tp)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if len(sys.argv) > 1 and sys.argv[1] == "one":
    os.environ["CASMTR_CQ_DYNAMIC"] = sys.argv[3]
    tp = ops.window_warp_idx(field(sys.argv[2]), hc, wc, 5)
    for _ in range(3):
        run(tp)
    torch.cuda.synchronize()
    sys.exit(0)

for name in ("smooth", "mix6", "mix12", "mix25", "random"):
    tp = ops.window_warp_idx(field(name), hc, wc, 5)
    ref = None
    for order in ("c", "r"):
        os.environ["CASMTR_CQ_ORDER"] = order
        for dyn in ("0", "1"):
            os.environ["CASMTR_CQ_DYNAMIC"] = dyn
            m = run(tp)
            torch.cuda.synchronize()
            if ref is None:
                ref = m
            same = torch.equal(ref, m)
            print(f"{name:7s} order {order} dynamic {dyn}: {timeit(tp):7.1f} us per launch   bit-equal to the first variant: {same}", flush=True)
    os.environ["CASMTR_CQ_ORDER"] = "r"
    os.environ["CASMTR_CQ_DYNAMIC"] = "1"
    for claim in (1, 2, 4, 8):
        os.environ["CASMTR_CQ_CLAIM"] = str(claim)
        print(f"{name:7s} order r dynamic 1, {claim} item(s) per claim: {timeit(tp):7.1f} us per launch   bit-equal: {torch.equal(ref, run(tp))}", flush=True)
    os.environ.pop("CASMTR_CQ_CLAIM", None)
    os.environ.pop("CASMTR_CQ_ORDER", None)
