"""Round 6: the persistent gather kernels (fine_quad, cascade_quad, window_match_pair) claim their items from per-XCD work counters
(csrc/common.hpp work_claim_issue, csrc/prof.hip work_counters) instead of a static stride.  The schedule must not be visible in the
results: every output bit-equal to the static schedule's (CASMTR_{FQ,CQ,WP}_DYNAMIC=0), for any claim size, on shapes with fewer items
than waves, with launches of different kernels interleaved on two streams (counter slots rotate), and the counters must be zero again
after every launch (they reset themselves; a stale value would make the next user of the slot skip items)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fine_case(g, B, side, Kp, topk, H=8):
    from casmtr_amd import ops
    C = 32 * H
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
    q, k, v = rn(B, side * side, C), rn(B, side * side, C), rn(B, side * side, C)
    Lq = (side // 2) ** 2
    prev = torch.stack([torch.argsort(torch.rand(B, Lq, Lq, generator=g, device=DEV), dim=-1)[..., :Kp] for _ in range(H)], -1).contiguous()
    acc = rn(B, Lq, C)
    hw = (side, side)
    args = (ops.tokens_to_quads(q, *hw), ops.tokens_to_quads(k, *hw), ops.tokens_to_quads(v, *hw), ops.topk_idx_to_tab(prev), hw, hw, H, topk)
    return lambda: ops.qta_fine_level_quad(*args, w_level=0.3, acc_in=acc, want_message=True, want_topk=topk > 0)


def _cascade_case(g, B, hc, wc, H=4, frac=0.2):
    from casmtr_amd import ops
    h, w = 2 * hc, 2 * wc
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
    q, k, v = (ops.tokens_to_quads(rn(B, h * w, 32 * H), h, w) for _ in range(3))
    ys, xs = torch.meshgrid(torch.arange(hc, device=DEV), torch.arange(wc, device=DEV), indexing="ij")
    ci = ((ys + 2).clamp(max=hc - 1) * wc + (xs + 3).clamp(max=wc - 1)).reshape(1, -1).repeat(B, 1)
    pick = torch.rand(ci.shape, generator=g, device=DEV) < frac
    ci = torch.where(pick, torch.randint(0, hc * wc, ci.shape, generator=g, device=DEV), ci)
    tp = ops.window_warp_idx(ci, hc, wc, 5)
    return lambda: {"message": ops.cascade_attn_quad(q, k, v, tp, (h, w), (h, w), H)}


def _window_case(g, B, hc, wc, C=128, frac=0.2):
    from casmtr_amd import ops
    h, w = 2 * hc, 2 * wc
    rn = lambda *s: torch.randn(*s, generator=g, device=DEV)
    fq, fk = rn(B, h * w, C), rn(B, h * w, C)
    ys, xs = torch.meshgrid(torch.arange(hc, device=DEV), torch.arange(wc, device=DEV), indexing="ij")
    ci = ((ys + 2).clamp(max=hc - 1) * wc + (xs + 3).clamp(max=wc - 1)).reshape(1, -1).repeat(B, 1)
    pick = torch.rand(ci.shape, generator=g, device=DEV) < frac
    ci = torch.where(pick, torch.randint(0, hc * wc, ci.shape, generator=g, device=DEV), ci)
    wi = ops.WindowIndex(ops.window_warp_idx(ci, hc, wc, 5), (h, w), (h, w), 1)
    return lambda: ops.window_match(fq, fk, wi, 1.0, recip=True, want_conf=True)


def _same(a, b):
    return all(torch.equal(a[k], b[k]) for k in a if torch.is_tensor(a[k]))


CASES = {
    "fine level 0 (lists of 64)": (lambda g: _fine_case(g, 3, 52, 16, 0), "FQ"),
    "fine level 1 (lists of 128, top-16)": (lambda g: _fine_case(g, 3, 28, 32, 16), "FQ"),
    "fine, fewer items than waves": (lambda g: _fine_case(g, 1, 8, 4, 2, H=2), "FQ"),
    "cascade 52 x 52 quads": (lambda g: _cascade_case(g, 3, 26, 26), "CQ"),
    "cascade, odd quad count per row": (lambda g: _cascade_case(g, 2, 15, 13, H=2), "CQ"),
    "cascade, a handful of items": (lambda g: _cascade_case(g, 1, 5, 5, H=1), "CQ"),
    "window match 52 x 52": (lambda g: _window_case(g, 3, 26, 26), "WP"),
    "window match C = 64, tiny": (lambda g: _window_case(g, 1, 6, 5, C=64), "WP"),
}


@pytest.mark.parametrize("name", list(CASES))
def test_dynamic_schedule_is_invisible(name, monkeypatch):
    from casmtr_amd import _lib
    make, tag = CASES[name]
    run = make(torch.Generator(device=DEV).manual_seed(3))
    monkeypatch.setenv(f"CASMTR_{tag}_DYNAMIC", "0")
    ref = run()
    torch.cuda.synchronize()
    monkeypatch.setenv(f"CASMTR_{tag}_DYNAMIC", "1")
    for claim in ("1", "2", "3", "4", "8", "64"):
        monkeypatch.setenv(f"CASMTR_{tag}_CLAIM", claim)
        for rep in range(3):
            out = run()
            assert _same(ref, out), f"{name}: {claim} item(s) per claim, repetition {rep}"
    assert _lib.lib().casmtr_debug_work_counters_nonzero() == 0, "every launch leaves its counter slot zeroed"


def test_interleaved_streams_share_no_counters():
    """200 launches of the three kernels alternating on two streams (more than the 64 counter slots of the device, so slots are reused
    while other launches are in flight on the other stream): every result equals its serial reference, counters end at zero"""
    from casmtr_amd import _lib
    g = torch.Generator(device=DEV).manual_seed(11)
    runs = [_fine_case(g, 2, 52, 16, 0), _cascade_case(g, 2, 26, 26), _window_case(g, 2, 26, 26), _fine_case(g, 2, 28, 32, 16)]
    refs = [r() for r in runs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    outs = []
    for i in range(200):
        with torch.cuda.stream(streams[i % 2]):
            outs.append((i % len(runs), runs[i % len(runs)]()))
    torch.cuda.synchronize()
    for k, o in outs:
        assert _same(refs[k], o), f"launch of case {k} on a shared device"
    assert _lib.lib().casmtr_debug_work_counters_nonzero() == 0
