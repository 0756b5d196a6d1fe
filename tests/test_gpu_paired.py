"""The two directions of an attention layer on shared launches (QTAttB.forward_multi / CascadeQTAttB.forward_multi, grouped layout pass):
results must equal the separate calls bit for bit -- same kernels, same per-item arithmetic, only the batch index differs.
Independence in the reference: src/model/modules/transformer.py:295-300 (both directions computed from the same inputs), :549."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pyr(g, B, C, h, w, levels=3):
    x = torch.randn((B, C, h, w), generator=g, device=DEV)
    out = [x]
    for _ in range(levels - 1):
        x = torch.nn.functional.avg_pool2d(x, 2, 2)
        out.append(x)
    return out


def test_qtattb_forward_multi_equals_separate_calls():
    from casmtr_amd.modules.quadtree_attention import QTAttB
    g = torch.Generator(device=DEV).manual_seed(0)
    att = QTAttB(8, 32, scale=3, topks=[32, 16, 8]).to(DEV).eval()
    with torch.no_grad():
        att.weight.copy_(torch.randn(3, generator=g, device=DEV))
        calls = [(_pyr(g, 2, 256, 40, 32), _pyr(g, 2, 256, 40, 32), _pyr(g, 2, 256, 40, 32)) for _ in range(2)]
        sep = [att(*c) for c in calls]
        both = att.forward_multi(calls)
        hybrid = att.forward_multi(calls, split_fine=True)   # shared layout + coarsest level, finer levels per call
    for a, b, c in zip(sep, both, hybrid):
        assert torch.equal(a, b) and torch.equal(a, c)


def test_cascade_forward_multi_equals_separate_calls():
    from casmtr_amd import ops
    from casmtr_amd.modules.quadtree_attention import CascadeQTAttB
    g = torch.Generator(device=DEV).manual_seed(1)
    att = CascadeQTAttB(4, 32, dilated=1).to(DEV).eval()
    B, h, w = 2, 32, 48
    hc, wc = h // 2, w // 2
    calls = []
    for _ in range(2):
        q, k, v = (torch.randn((B, 128, h, w), generator=g, device=DEV) for _ in range(3))
        idx = torch.randint(0, hc * wc, (B, hc * wc), generator=g, device=DEV)
        calls.append((q, k, v, ops.window_warp_idx(idx, hc, wc, 5)))
    with torch.no_grad():
        sep = [att(q, k, v, tp, None, want_idx=False)[0] for q, k, v, tp in calls]
        both = att.forward_multi(calls)
        hybrid = att.forward_multi(calls, split_attn=True)
    for a, b, c in zip(sep, both, hybrid):
        assert torch.equal(a, b) and torch.equal(a, c)


def test_grouped_layout_equals_single():
    from casmtr_amd import ops
    g = torch.Generator(device=DEV).manual_seed(2)
    xs = [[torch.randn((3, 64, 12, 20), generator=g, device=DEV) for _ in range(2)] for _ in range(4)]
    outs = ops.nchw_to_quads_grouped(xs, [True, False, False, True])
    for grp, o, tok in zip(xs, outs, [True, False, False, True]):
        ref = ops.nchw_to_quads_multi(grp, [tok] * 2)
        assert torch.equal(o[:3], ref[0]) and torch.equal(o[3:], ref[1])
