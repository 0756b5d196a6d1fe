"""PostProcess methods outside the shipped configs (src/model/functions/post_processing.py).

CPU: the oracle's restatement of kornia 0.6.2 conv_soft_argmax2d on cases worked out by hand (parity with kornia itself is
UNPINNED: kornia is neither in /root/reference nor in this image), the reference's error behaviour for stride != 1 and 'sift'.
GPU: PostProcess.apply for 'softargmax_nms' and 'd2d' against the oracle; 'd2d' is also pinned through the reference-generated
fixture cascade_matching_d2d (tests/test_oracle_golden.py, tests/test_gpu_ops.py, tests/test_gpu_pipeline.py)."""
import numpy as np
import pytest
import torch

import oracle


def test_soft_argmax_hand_cases():
    # one dominant peak: every window that contains it is pulled towards it by at most one pixel (offsets are in kornia's
    # normalised window units, -1 .. 1), windows that do not contain it answer their own centre
    h = w = 7
    x = np.zeros((1, h * w), np.float32)
    x[0, 3 * w + 3] = 50.0
    keep = oracle.conv_soft_argmax_mask(x, (h, w), 3).reshape(h, w)
    want = np.ones((h, w), bool)
    # 3x3 windows centred on the 8 neighbours of (3,3) land ON (3,3) (offset -1/+1 towards it) -> the neighbours themselves are never hit
    want[2:5, 2:5] = False
    want[3, 3] = True
    assert np.array_equal(keep, want)
    # uniform input: interior windows answer their centre; at the border the zero padding pulls the mean half a pixel inwards:
    # 0 + 0.5 rounds (half to even, torch.round and np.rint alike) to 0, 5 - 0.5 = 4.5 to 4 -> the last row / column is never hit
    keep = oracle.conv_soft_argmax_mask(np.ones((2, 36), np.float32), (6, 6), 3).reshape(2, 6, 6)
    want = np.zeros((6, 6), bool)
    want[:5, :5] = True
    assert np.array_equal(keep[0], want) and np.array_equal(keep[1], want)
    # non-overlapping windows (stride == window): one position per window
    keep = oracle.conv_soft_argmax_mask(np.random.default_rng(0).random((1, 64), dtype=np.float32), (8, 8), 4, stride=4)
    assert keep.sum() == 4


def test_reference_error_behaviour():
    from casmtr_amd.matching.post_processing import PostProcess
    with pytest.raises(NotImplementedError):
        PostProcess({"method": "sift"})
    with pytest.raises(NotImplementedError):
        PostProcess({"method": "nope"})
    for m in ("maxpool_nms", "d2d"):   # pooled index map vs full coordinate map: broadcast error in the reference (:118-119, :129-130)
        pp = PostProcess({"method": m, "window_size": 5, "stride": 5})
        with pytest.raises(RuntimeError, match="must match the size"):
            pp.extra_mask(torch.zeros((1, 100)), (10, 10), data={})


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(window_size=5), dict(window_size=3, temperature=0.05), dict(window_size=4, stride=4, temperature=0.1)])
def test_softargmax_nms_vs_oracle(cfg):
    from casmtr_amd.matching.post_processing import PostProcess
    r = np.random.default_rng(7)
    B, h, w = 2, 16, 16            # square: the reference's channel0 * w0c + channel1 stays in range
    conf = r.random((B, h * w), dtype=np.float32)
    idx = r.integers(0, h * w, (B, h * w), dtype=np.int64)
    pp = PostProcess(dict(method="softargmax_nms", **cfg))
    axes = {"h0c": h, "w0c": w, "h1c": h, "w1c": w}
    got = pp.apply({}, axes, torch.from_numpy(idx).cuda(), torch.from_numpy(conf).cuda(), 0.3, "4c").cpu().numpy()
    want = oracle.conv_soft_argmax_mask(conf, (h, w), cfg["window_size"], cfg.get("stride", 1), cfg.get("temperature", 1.0)) & (conf > 0.3)
    # positions within float rounding of x.5 may round differently between the fp32 convolutions and the oracle's float64 sums
    assert (got != want).sum() <= 2, (got != want).sum()
    assert got.sum() > 10


@pytest.mark.gpu
def test_d2d_vs_oracle():
    from casmtr_amd.matching.cascade_matching import CascadeMatching
    from casmtr_amd.matching.post_processing import PostProcess
    r = np.random.default_rng(8)
    B, h, w, C = 2, 24, 32, 64
    feat = r.standard_normal((B, h * w, C), dtype=np.float32)
    conf = r.random((B, h * w), dtype=np.float32)
    s_gpu = CascadeMatching._d2d_scores(torch.from_numpy(feat).cuda(), (h, w)).cpu().numpy()
    s_ora = oracle.d2d_scores(feat, (h, w))
    assert np.abs(s_gpu - s_ora).max() < 1e-5
    pp = PostProcess(dict(method="d2d", window_size=5))
    data = {"S_d2d": torch.from_numpy(s_ora).cuda(), "d2d_w": w // 4}
    axes = {"h0c": h, "w0c": w, "h1c": h, "w1c": w}
    idx = torch.zeros((B, h * w), dtype=torch.int64).cuda()
    got = pp.apply(data, axes, idx, torch.from_numpy(conf).cuda(), 0.2, "4c").cpu().numpy()
    want = oracle.d2d_mask(conf, s_ora, (h, w), 5) & (conf > 0.2)
    assert np.array_equal(got, want) and want.sum() > 10
