"""Measurement variants of the finest QTAttB level (csrc/fine_lw.hip: loader-wave specialisation; csrc/fine_vs.hip: gathers in flight in
VGPRs; selected by CASMTR_FQ_VARIANT): bit-equal to the shipped fine_quad_kernel<1, false, true> on the same inputs
(QuadtreeAttention/modules/quadtree_attention.py:180-229 with 64-candidate lists and no top-k)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(B, H, side, with_acc, want_message, seed):
    from casmtr_amd import ops

    Kp, C = 16, 32 * H
    g = torch.Generator(device="cuda").manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
    hw = (side, side)
    q, k, v = rn(B, side * side, C), rn(B, side * side, C), rn(B, side * side, C)
    Lq = (side // 2) ** 2
    prev = torch.stack([torch.argsort(torch.rand(B, Lq, Lq, generator=g, device="cuda"), dim=-1)[..., :Kp] for _ in range(H)], -1).contiguous()
    acc = rn(B, Lq, C) if with_acc else None
    qq, kq, vq, tab = ops.tokens_to_quads(q, *hw), ops.tokens_to_quads(k, *hw), ops.tokens_to_quads(v, *hw), ops.topk_idx_to_tab(prev)

    def run():
        out = ops.qta_fine_level_quad(qq, kq, vq, tab, hw, hw, H, 0, w_level=0.3, acc_in=acc, want_message=want_message, want_topk=False)
        torch.cuda.synchronize()
        return out

    return run


@pytest.mark.parametrize("variant", ["lw", "vs"])
@pytest.mark.parametrize("B,H,side,with_acc,want_message", [(2, 8, 24, True, True), (3, 4, 16, True, False), (1, 2, 12, False, True),
                                                           (5, 8, 10, True, False), (1, 1, 8, True, True)])
def test_fine_variant_bit_equal(monkeypatch, variant, B, H, side, with_acc, want_message):
    run = _run(B, H, side, with_acc, want_message, seed=B * 100 + H * 10 + side)
    monkeypatch.delenv("CASMTR_FQ_VARIANT", raising=False)
    ref = run()
    monkeypatch.setenv("CASMTR_FQ_VARIANT", variant)
    for _ in range(3):   # persistent grids with LDS hand-shakes: more than one launch
        poison = [torch.full_like(ref[kk], float("nan")) for kk in ("acc", "message") if ref.get(kk) is not None]   # the allocator reuses these
        del poison
        out = run()
        for kk in ("acc", "message"):
            assert (ref.get(kk) is None) == (out.get(kk) is None)
            if ref.get(kk) is not None:
                assert torch.equal(out[kk], ref[kk]), (variant, kk)
    assert os.environ["CASMTR_FQ_VARIANT"] == variant
