"""CPU: the oracle (oracle/casmtr_oracle.c) against fixtures produced by the reference python itself.

This is what pins the oracle (the reference has no tests of its own).  Index outputs must bit-match;
a mismatch is only tolerated when the audit shows a genuine near tie (the reference selects on softmax values with
torch.topk / torch.max, the canonical arithmetic selects on logits -- SURVEY.md §7 'Hard parts').
"""
import numpy as np
import pytest

import oracle
from golden_inputs import CASES, checksum, make_inputs
from parity_utils import assert_close, audit_index_mismatches, dot_score_fn, load_golden, match_set, post_extra_mask

F32_SUM_TOL = 5e-5      # different fp32 summation order (torch.sum / einsum vs fmaf chain)
SOFTMAX_TOL = 1e-4      # north_star: softmax scores within 1e-4 fp32


def _inputs(group, name):
    inp = make_inputs(group, name)
    g = load_golden(group, name)
    assert (int(g["checksum"][0]) & 0xFFFFFFFF) == int(checksum(inp)[0]), "seeded inputs drifted from the ones the fixture was made with"
    return inp, g


@pytest.mark.parametrize("name", list(CASES["ops"]))
def test_ops(name):
    inp, g = _inputs("ops", name)
    s = oracle.qta_score_fwd(inp["q"], inp["key"], inp["idx"])
    assert_close(s, g["score"], F32_SUM_TOL, "qta_score_fwd")
    A = g["agg_in_score"]  # [B,N1,4,K,H]
    B, N1, _, K, H = A.shape
    idx5 = np.repeat(inp["idx"][:, :, None], 4, axis=2)
    m = oracle.qta_value_agg_fwd(A.reshape(B, N1 * 4, K, H), inp["value"], idx5.reshape(B, N1 * 4, K, H))
    assert_close(m.reshape(g["message"].shape), g["message"], 1e-5, "qta_value_agg_fwd")
    ws = oracle.window_score_fwd(inp["wq"], inp["wkey"], inp["widx"])
    assert_close(ws, g["window_score"], F32_SUM_TOL, "window_score_fwd")


def _audit_topk(idx_o, idx_g, score_o):
    """idx_*: [B,L,k,H] ; score_o: oracle's softmax score at idx_o.  Returns number of mismatching series; asserts each
    mismatch is a near tie (same set up to elements whose oracle scores are within 1e-6 relative of the k-th score)."""
    bad = np.argwhere((idx_o != idx_g).any(axis=2))
    for b, l, h in bad:
        so, sg = set(idx_o[b, l, :, h].tolist()), set(idx_g[b, l, :, h].tolist())
        kth = score_o[b, l, -1, h]
        # elements chosen by one side only must sit at the k-th boundary
        only_o = [i for i in idx_o[b, l, :, h] if i not in sg]
        for i in only_o:
            pos = idx_o[b, l, :, h].tolist().index(i)
            assert abs(score_o[b, l, pos, h] - kth) <= 1e-6 * max(abs(kth), 1e-30) + 1e-12, "top-k set differs beyond a near tie"
        assert len(so - sg) == len(sg - so)
    return len(bad)


@pytest.mark.parametrize("name", list(CASES["qtattb"]))
def test_qtattb(name):
    inp, g = _inputs("qtattb", name)
    cfg = CASES["qtattb"][name]
    final, levels = oracle.qtattb_forward(inp["queries"], inp["keys"], inp["values"], inp["weight"], cfg["nhead"], cfg["topks"])
    n_series = n_bad = 0
    for lv, out in enumerate(levels):
        assert bool(g[f"L{lv}_smart_idx_equal"][0]), "reference's own two paths disagree on this fixture"
        gi = g[f"L{lv}_topk_idx"].astype(np.int64)
        n_bad += _audit_topk(out["topk_idx"], gi, out["topk_score"])
        n_series += gi.shape[0] * gi.shape[1] * gi.shape[3]
        if f"L{lv}_topk_score" in g:
            if n_bad == 0:
                assert_close(out["topk_score"], g[f"L{lv}_topk_score"], SOFTMAX_TOL, f"L{lv} topk_score")
    if n_bad == 0:  # a near-tie flip legitimately changes every finer level below it
        assert_close(final, g["final"], SOFTMAX_TOL, "final message")
        if "L0_message" in g:
            assert_close(levels[0]["message"], g["L0_message"], SOFTMAX_TOL, "L0 message")
            for lv in (1, 2):
                gm = g[f"L{lv}_message"]  # [B,L/4,4,H,D] quad order
                B, Lq, _, H, D = gm.shape
                hq = inp["queries"][2 - lv].shape[2] // 2
                wq = Lq // hq
                m = levels[lv]["message"].reshape(B, hq, 2, wq, 2, H, D).transpose(0, 1, 3, 2, 4, 5, 6).reshape(B, Lq, 4, H, D)
                assert_close(m, gm, SOFTMAX_TOL, f"L{lv} message")
    assert n_bad <= max(1, n_series // 5000), f"{n_bad}/{n_series} top-k series differ from the reference"
    assert float(g["smart_vs_cuda_maxabs"][0]) < 1e-5


@pytest.mark.parametrize("name", list(CASES["cascade_attn"]))
def test_cascade_attn(name):
    inp, g = _inputs("cascade_attn", name)
    cfg = CASES["cascade_attn"][name]
    hc, wc = cfg["coarse_hw"]
    tp = oracle.window_warp_idx(inp["coarse_idx"], hc, wc, cfg["ws"])
    assert np.array_equal(tp, g["topk_pos"].astype(np.int64)), "window_warp_idx"
    B, C, h, w = inp["q"].shape
    tok = lambda x: np.ascontiguousarray(x.transpose(0, 2, 3, 1).reshape(B, h * w, C))
    msg, up = oracle.cascade_attn(tok(inp["q"]), tok(inp["k"]), tok(inp["v"]), tp, (h, w), (h, w), cfg["nhead"],
                                  rel_pos=inp.get("rel_pos"))
    assert np.array_equal(up, g["upsampled_idx"].astype(np.int64)), "upsampled_idx"
    assert_close(msg, g["message"], SOFTMAX_TOL, "message")


@pytest.mark.parametrize("name", list(CASES["quadtree_block"]))
def test_quadtree_block(name):
    """§8 f.1 callers: projections + pyramid + attention + output projection against the reference's modules."""
    inp, g = _inputs("quadtree_block", name)
    cfg = CASES["quadtree_block"][name]
    if cfg["kind"] == "qta":
        hw, hw1 = cfg["hw"], cfg.get("hw1", cfg["hw"])
        out, levels = oracle.quadtree_attention_block(
            inp["x"], inp["target"], hw, hw1, inp["wq"], inp["wk"], inp["wv"], inp["weight"], inp["wp"], inp["bp"],
            cfg["nhead"], cfg["topks"], 3, inp.get("bq"), inp.get("bk"), inp.get("bv"))
        n_bad = 0
        for lv, o in enumerate(levels):
            n_bad += _audit_topk(o["topk_idx"], g[f"L{lv}_topk_idx"].astype(np.int64), o["topk_score"])
        if cfg["exact"]:  # q/k/v are exact in both implementations: the attention sees identical inputs
            assert n_bad == 0
        if n_bad == 0:
            assert_close(out, g["out"], SOFTMAX_TOL, "block output")
    else:
        hc, wc = cfg["coarse_hw"]
        tp = oracle.window_warp_idx(inp["coarse_idx"], hc, wc, cfg["ws"])
        assert np.array_equal(tp, g["topk_pos"].astype(np.int64))
        out, up = oracle.cascade_quadtree_attention_block(
            inp["x"], inp["target"], (2 * hc, 2 * wc), (2 * hc, 2 * wc), tp, inp["wq"], inp["wk"], inp["wv"], inp["wp"],
            inp["bp"], cfg["nhead"])
        assert np.array_equal(up, g["upsampled_idx"].astype(np.int64))
        assert_close(out, g["out"], SOFTMAX_TOL, "block output")


def test_token_pool_is_torch_avg_pool2d():
    """the pooling order is ATen's: bit-exact against F.avg_pool2d on CPU, odd sizes included"""
    import torch
    import torch.nn.functional as F
    r = np.random.default_rng(5)
    for (h, w, c) in ((12, 10, 16), (7, 9, 8)):
        x = r.standard_normal((2, h * w, c)).astype(np.float32)
        t = torch.from_numpy(x).view(2, h, w, c).permute(0, 3, 1, 2).contiguous()
        ref = F.avg_pool2d(t, kernel_size=2, stride=2).permute(0, 2, 3, 1).reshape(2, -1, c).numpy()
        assert np.array_equal(oracle.token_pool(x, h, w), ref)


def _check_matches(o, g, conf_tol_ok):
    so, sg = match_set(o["b_ids"], o["i_ids"], o["j_ids"]), match_set(g["b_ids"], g["i_ids"], g["j_ids"])
    for extra in (so ^ sg):
        assert conf_tol_ok(extra), f"match list differs at {extra} and it is not a threshold / tie borderline"
    return len(so ^ sg)


@pytest.mark.parametrize("name", list(CASES["coarse_matching"]))
def test_coarse_matching(name):
    inp, g = _inputs("coarse_matching", name)
    cfg = CASES["coarse_matching"][name]
    valid = None
    if cfg.get("masks"):
        m0, m1 = inp["mask0"], inp["mask1"]
        valid = np.stack([m0.sum(1).max(-1), m0.sum(2).max(-1), m1.sum(1).max(-1), m1.sum(2).max(-1)], 1).astype(np.int32)
    o = oracle.dual_softmax(inp["feat0"], inp["feat1"], cfg["hw0"], cfg["hw1"], temperature=cfg.get("T", 0.1),
                            thr=cfg.get("thr", 0.2), border_rm=cfg.get("border_rm", 0),
                            mask0=inp["mask0"].reshape(cfg["B"], -1) if cfg.get("masks") else None,
                            mask1=inp["mask1"].reshape(cfg["B"], -1) if cfg.get("masks") else None,
                            valid_hw=valid, recip=False, want_conf=True)
    # argmax indices must equal the reference's; a difference is accepted only as an audited near tie of the similarity
    # (fully masked rows are uniform -> index 0 on both sides)
    mk0 = inp["mask0"].reshape(cfg["B"], -1) if cfg.get("masks") else None
    mk1 = inp["mask1"].reshape(cfg["B"], -1) if cfg.get("masks") else None
    audit_index_mismatches(o["next_idx_c01"], g["next_idx_c01"], dot_score_fn(inp["feat0"], inp["feat1"], mk0, mk1), "next_idx_c01")
    audit_index_mismatches(o["next_idx_c10"], g["next_idx_c10"], dot_score_fn(inp["feat1"], inp["feat0"], mk1, mk0), "next_idx_c10")
    assert_close(o["next_conf_c01"], g["next_conf_c01"], SOFTMAX_TOL, "next_conf_c01")
    assert_close(o["next_conf_c10"], g["next_conf_c10"], SOFTMAX_TOL, "next_conf_c10")
    if "conf_matrix" in g:
        assert_close(o["conf_matrix"], g["conf_matrix"], SOFTMAX_TOL, "conf_matrix")
    assert_close(o["conf_matrix"].max(2), g["conf_rowmax"], SOFTMAX_TOL, "conf rowmax")
    assert_close(o["conf_matrix"].max(1), g["conf_colmax"], SOFTMAX_TOL, "conf colmax")
    thr = cfg.get("thr", 0.2)
    ndiff = _check_matches(o, g, lambda t: abs(o["conf_matrix"][t[0], t[1], t[2]] - thr) < 1e-4)
    assert ndiff <= 1
    assert len(g["b_ids"]) > 10, "fixture should contain confident matches"
    if ndiff == 0:
        assert_close(o["mconf"], g["mconf"], SOFTMAX_TOL, "mconf")


@pytest.mark.parametrize("name", list(CASES["cascade_matching"]))
def test_cascade_matching(name):
    inp, g = _inputs("cascade_matching", name)
    cfg = CASES["cascade_matching"][name]
    hc, wc = cfg["coarse_hw"]
    h, w = 2 * hc, 2 * wc
    B = cfg["B"]
    # window index lists, built the way the model builds them (window -> children), checked against the fixture
    idx = {}
    for key, gk in (("coarse_idx01", "idx_c01"), ("coarse_idx10", "idx_c10")):
        tp = oracle.window_warp_idx(inp[key], hc, wc, 5)
        z = np.zeros((B, h * w, 128), np.float32)
        _, up = oracle.cascade_attn(z, z, z, tp, (h, w), (h, w), 4)
        assert np.array_equal(up, g[gk].astype(np.int64))
        idx[gk] = up
    mq = mk = None
    valid = None
    if cfg.get("masks"):
        m0, m1 = inp["mask0"], inp["mask1"]
        mq, mk = m0.reshape(B, -1), m1.reshape(B, -1)
        valid = np.stack([m0.sum(1).max(-1), m0.sum(2).max(-1), m1.sum(1).max(-1), m1.sum(2).max(-1)], 1).astype(np.int32)
    d01 = oracle.window_match(inp["feat0"], inp["feat1"], idx["idx_c01"], 1.0, mq, mk)
    d10 = oracle.window_match(inp["feat1"], inp["feat0"], idx["idx_c10"], 1.0, mk, mq)
    assert_close(d01["conf_matrix"], g["conf_matrix"], SOFTMAX_TOL, "conf_matrix01")
    assert_close(d01["next_conf"], g["next_conf_c01"], SOFTMAX_TOL, "next_conf_c01")
    assert_close(d10["next_conf"], g["next_conf_c10"], SOFTMAX_TOL, "next_conf_c10")
    # the reference takes argmax of softmax VALUES (ties -> first), the oracle argmax of logits: equal, or an audited near tie
    audit_index_mismatches(d01["next_idx"], g["next_idx_c01"], dot_score_fn(inp["feat0"], inp["feat1"], mq, mk), "cascade next_idx_c01")
    audit_index_mismatches(d10["next_idx"], g["next_idx_c10"], dot_score_fn(inp["feat1"], inp["feat0"], mk, mq), "cascade next_idx_c10")
    post, extra = cfg.get("post"), None
    if post:   # PostProcess 'local_window_nms' / 'd2d' (post_processing.py:76-93, :122-143), restated in the oracle
        extra = post_extra_mask(post, g["next_conf_c01"], inp["feat0"], (h, w))
    sel = oracle.nms_select(g["next_conf_c01"], g["next_idx_c01"].astype(np.int64), g["next_idx_c10"].astype(np.int64),
                            (h, w), (h, w), nms_window=5 if (cfg.get("nms", True) and not post) else 0, extra_keep=extra,
                            test_thr=cfg.get("test_thr", 0.2), pre=[(inp["pre_conf"], (hc, wc), cfg.get("pre_thr", 0.2))],
                            border_rm=cfg.get("border_rm", 2), valid_hw=valid, double_check=cfg.get("double_check", True))
    # fed with the reference's own stage outputs the selection is pure integer / comparison logic: exact
    assert np.array_equal(sel["b_ids"], g["b_ids"].astype(np.int64))
    assert np.array_equal(sel["i_ids"], g["i_ids"].astype(np.int64))
    assert np.array_equal(sel["j_ids"], g["j_ids"].astype(np.int64))
    assert np.array_equal(sel["mconf"], g["mconf"])
    assert len(g["b_ids"]) > 5
    scale = 4.0
    mk0 = np.stack([sel["i_ids"] % w, sel["i_ids"] // w], 1) * scale
    assert np.array_equal(mk0.astype(np.float32), g["mkpts0_c"].astype(np.float32))
