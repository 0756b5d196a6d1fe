"""Static item schedule against dynamic claiming (round 6) for the persistent gather kernels at the CasMTR-4c bench shapes (B = 8):
fine level 0 (104 x 104, lists of 64), fine level 1 (52 x 52, lists of 128, top-16), window matching (208 x 208, C = 128, 12 % random
windows).  us per launch, bit-equality of every output, and the wave-count sweep of the fine levels under both schedules."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import ops

g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def same(a, b):
    return all(torch.equal(a[k], b[k]) for k in a if torch.is_tensor(a[k]) and torch.is_tensor(b.get(k)))


B, H, C = 8, 8, 256
for level, (side, Kp, topk) in enumerate(((104, 16, 0), (52, 32, 16))):
    hw = (side, side)
    q, k, v = rn(B, side * side, C), rn(B, side * side, C), rn(B, side * side, C)
    Lq = (side // 2) ** 2
    prev = torch.stack([torch.argsort(torch.rand(B, Lq, Lq, generator=g, device="cuda"), dim=-1)[..., :Kp] for _ in range(H)], -1).contiguous()
    acc = rn(B, Lq, C)
    qq, kq, vq, tab = ops.tokens_to_quads(q, *hw), ops.tokens_to_quads(k, *hw), ops.tokens_to_quads(v, *hw), ops.topk_idx_to_tab(prev)
    run = lambda: ops.qta_fine_level_quad(qq, kq, vq, tab, hw, hw, H, topk, w_level=0.3, acc_in=acc, want_message=True, want_topk=topk > 0)
    ref = None
    for wpx in (None, 256, 320, 384, 448, 512):
        if wpx is None:
            os.environ.pop("CASMTR_FQ_WAVES_PER_XCD", None)
        else:
            os.environ["CASMTR_FQ_WAVES_PER_XCD"] = str(wpx)
        for dyn, claim in (("0", 1), ("1", 1), ("1", 2), ("1", 4), ("1", 8), ("1", 16)):
            os.environ["CASMTR_FQ_DYNAMIC"] = dyn
            os.environ["CASMTR_FQ_CLAIM"] = str(claim)
            out = run()
            torch.cuda.synchronize()
            ref = out if ref is None else ref
            print(f"fine level {level}: waves per XCD {wpx or 'default'}, dynamic {dyn}, {claim} item(s) per claim: {timeit(run):7.1f} us per launch   bit-equal: {same(ref, out)}", flush=True)
    os.environ.pop("CASMTR_FQ_WAVES_PER_XCD", None)
    os.environ.pop("CASMTR_FQ_CLAIM", None)

B, C, hc, wc = 8, 128, 104, 104
h, w = 2 * hc, 2 * wc
fq, fk = rn(B, h * w, C), rn(B, h * w, C)
ys, xs = torch.meshgrid(torch.arange(hc, device="cuda"), torch.arange(wc, device="cuda"), indexing="ij")
cidx = ((ys + 3).clamp(max=hc - 1) * wc + (xs + 5).clamp(max=wc - 1)).reshape(1, -1).repeat(B, 1)
for frac in (0.0, 0.12):
    pick = torch.rand(cidx.shape, generator=g, device="cuda") < frac
    ci = torch.where(pick, torch.randint(0, hc * wc, cidx.shape, generator=g, device="cuda"), cidx)
    wi = ops.WindowIndex(ops.window_warp_idx(ci, hc, wc, 5), (h, w), (h, w), 1)
    run = lambda: ops.window_match(fq, fk, wi, 1.0, recip=True, want_conf=True)
    ref = None
    for dyn, claim in (("0", 1), ("1", 1), ("1", 2), ("1", 4)):
        os.environ["CASMTR_WP_DYNAMIC"] = dyn
        os.environ["CASMTR_WP_CLAIM"] = str(claim)
        out = run()
        torch.cuda.synchronize()
        ref = out if ref is None else ref
        eq = all(torch.equal(a, b) for a, b in zip(ref, out)) if isinstance(out, (tuple, list)) else same(ref, out)
        print(f"window match, {int(frac * 100)} % random windows, dynamic {dyn}, {claim} item(s) per claim: {timeit(run):7.1f} us per launch   bit-equal: {eq}", flush=True)
