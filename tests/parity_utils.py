"""Helpers shared by the CPU (oracle vs golden) and GPU (HIP vs oracle) parity tests."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(group, name):
    z = np.load(os.path.join(GOLD, f"{group}_{name}.npz"))
    return {k: z[k] for k in z.files}


def assert_close(a, b, atol, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= atol, f"{what}: max abs err {err:.3e} > {atol:.1e}"
    return err


def near_tie(sa, sb, rel=4e-6, absolute=1e-6):
    """two fp32-chain scores that an exact (float64) evaluation cannot tell apart beyond accumulation-order rounding"""
    return abs(sa - sb) <= absolute + rel * max(abs(sa), abs(sb))


def audit_index_mismatches(got, want, score_fn, what):
    """Index tensors (argmax / gathered absolute indices) must be EQUAL; an element may differ only as a NEAR TIE:
    `score_fn(position_tuple, index) -> float` evaluates, in float64, the quantity the index was selected on, and the two
    candidates' scores must agree within fp32 rounding.  Returns the number of (audited) mismatches."""
    got, want = np.asarray(got).astype(np.int64), np.asarray(want).astype(np.int64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    bad = np.argwhere(got != want)
    for pos in bad:
        pos = tuple(int(x) for x in pos)
        sg, sw = float(score_fn(pos, int(got[pos]))), float(score_fn(pos, int(want[pos])))
        assert near_tie(sg, sw), f"{what}: index differs at {pos}: {got[pos]} (score {sg!r}) vs {want[pos]} ({sw!r}) -- not a near tie"
    return len(bad)


def audit_topk_mismatches(got, want, logit_fn, what):
    """Top-k index tensors [B, L, k, H] (sorted by score, descending) must be EQUAL; a (b, l, h) series may differ only by
    near ties: `logit_fn(b, l, h, idx_vector) -> float64 logits` of the selected keys; rank by rank the two lists must carry
    (numerically) the same logit, i.e. they differ by swaps of near-equal scores or at the k-th boundary by a near-equal
    substitute.  Returns the number of (audited) mismatching series."""
    got, want = np.asarray(got).astype(np.int64), np.asarray(want).astype(np.int64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    bad = np.argwhere((got != want).any(axis=2))
    for b, l, h in bad:
        lg, lw = logit_fn(b, l, h, got[b, l, :, h]), logit_fn(b, l, h, want[b, l, :, h])
        for t, (x, y) in enumerate(zip(lg, lw)):
            assert near_tie(float(x), float(y)), f"{what}: series (b={b}, l={l}, h={h}) differs beyond a near tie at rank {t}: {x!r} vs {y!r}"
    return len(bad)


def dot_score_fn(fa, fb, mask_a=None, mask_b=None):
    """score_fn for audit_index_mismatches on similarity argmaxes: position (b, n) with candidate index j -> <fa[b,n], fb[b,j]>
    in float64 (-1e9 where either side is masked, as the reference fills).  Scale factors are monotone and irrelevant."""
    fa, fb = np.asarray(fa, np.float64), np.asarray(fb, np.float64)

    def fn(pos, j):
        b, n = pos
        if mask_a is not None and not (mask_a[b, n] and mask_b[b, j]):
            return -1e9
        return float(fa[b, n] @ fb[b, j])
    return fn


def match_set(b, i, j):
    return set(zip(np.asarray(b).tolist(), np.asarray(i).tolist(), np.asarray(j).tolist()))


def post_extra_mask(post, next_conf_c01, feat0, hw):
    """the oracle's restatement of the PostProcess methods that reach the selection as an extra keep mask:
    'local_window_nms' (post_processing.py:76-93), 'd2d' (:122-143 + cascade_matching.py:88-104), 'softargmax_nms' (:93-110)"""
    import oracle
    if not post or post["method"] in (None, "maxpool_nms"):
        return None
    if post["method"] == "local_window_nms":
        return oracle.local_window_topk_mask(next_conf_c01, hw, post["window_size"], post["topk"])
    if post["method"] == "d2d":
        return oracle.d2d_mask(next_conf_c01, oracle.d2d_scores(feat0, hw), hw, post["window_size"])
    if post["method"] == "softargmax_nms":
        return oracle.conv_soft_argmax_mask(next_conf_c01, hw, post["window_size"], post.get("stride", 1), post.get("temperature", 1.0))
    raise KeyError(post["method"])
