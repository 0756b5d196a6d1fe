"""GPU: the module surface (QTAttB / CascadeQTAttB / CoarseMatching / CascadeMatching) against the reference-python
fixtures, and the whole hot-path chain at the BASELINE size against the oracle / size-independent properties."""
import numpy as np
import pytest
import torch

import oracle
from golden_inputs import CASES, make_inputs
from parity_utils import assert_close, audit_index_mismatches, dot_score_fn, load_golden, match_set

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("name", list(CASES["qtattb"]))
def test_qtattb_module_vs_reference(name):
    from casmtr_amd.modules.quadtree_attention import QTAttB
    inp = make_inputs("qtattb", name)
    cfg = CASES["qtattb"][name]
    g = load_golden("qtattb", name)
    m = QTAttB(cfg["nhead"], cfg["D"], scale=3, topks=cfg["topks"]).to(DEV).eval()
    with torch.no_grad():
        m.weight.copy_(T(inp["weight"]))
        out = m([T(x) for x in inp["queries"]], [T(x) for x in inp["keys"]], [T(x) for x in inp["values"]])
    assert out.shape == g["final"].shape
    assert_close(N(out), g["final"], TOL, "QTAttB fused forward vs reference python")
    # the differentiable composed path (primitives + torch) must agree with the fused one
    qs = [T(x).requires_grad_(True) for x in inp["queries"]]
    out2 = m(qs, [T(x) for x in inp["keys"]], [T(x) for x in inp["values"]])
    assert_close(N(out2), g["final"], TOL, "QTAttB composed forward vs reference python")
    out2.sum().backward()
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in qs)
    assert m.weight.grad is not None


@pytest.mark.parametrize("name", list(CASES["cascade_attn"]))
def test_cascade_module_vs_reference(name):
    from casmtr_amd.modules.quadtree_attention import CascadeQTAttB
    inp = make_inputs("cascade_attn", name)
    cfg = CASES["cascade_attn"][name]
    g = load_golden("cascade_attn", name)
    m = CascadeQTAttB(cfg["nhead"], cfg["D"], dilated=1).to(DEV)
    tp = T(g["topk_pos"].astype(np.int64))
    rel = T(inp["rel_pos"]) if cfg.get("rel_pos") else None
    with torch.no_grad():
        msg, up = m(T(inp["q"]), T(inp["k"]), T(inp["v"]), tp, rel)
    assert np.array_equal(N(up), g["upsampled_idx"].astype(np.int64))
    assert_close(N(msg), g["message"], TOL, "fused")
    q = T(inp["q"]).requires_grad_(True)
    msg2, up2 = m(q, T(inp["k"]), T(inp["v"]), tp, rel)
    assert np.array_equal(N(up2), g["upsampled_idx"].astype(np.int64))
    assert_close(N(msg2), g["message"], TOL, "composed")
    msg2.square().sum().backward()
    assert torch.isfinite(q.grad).all()


@pytest.mark.parametrize("gemm", [None, "split"])   # module default (exact fp32 chain for every logit) | opt-in f16 split + exact argmax re-decision
@pytest.mark.parametrize("name", list(CASES["coarse_matching"]))
def test_coarse_matching_module(name, gemm):
    from casmtr_amd.matching.coarse_matching import CoarseMatching
    inp = make_inputs("coarse_matching", name)
    cfg = CASES["coarse_matching"][name]
    g = load_golden("coarse_matching", name)
    mc = {"thr": cfg.get("thr", 0.2), "border_rm": cfg.get("border_rm", 0), "train_coarse_percent": 0.3,
          "train_pad_num_gt_min": 200, "match_type": "dual_softmax", "dsmax_temperature": cfg.get("T", 0.1)}
    cm = CoarseMatching(mc, div_mode="cpu", gemm=gemm).eval()   # fixtures come from the reference on CPU
    h0, w0 = cfg["hw0"]
    h1, w1 = cfg["hw1"]
    data = {"hw0_i": (h0 * 8, w0 * 8), "hw1_i": (h1 * 8, w1 * 8), "hw0_8c": (h0, w0), "hw1_8c": (h1, w1)}
    m0 = m1 = None
    if cfg.get("masks"):
        data["mask_8c0"], data["mask_8c1"] = T(inp["mask0"]).bool(), T(inp["mask1"]).bool()
        m0, m1 = data["mask_8c0"].flatten(-2), data["mask_8c1"].flatten(-2)
    with torch.no_grad():
        cm(T(inp["feat0"]), T(inp["feat1"]), data, mask_c0=m0, mask_c1=m1, level="8c")
    st = data["stage_8c"]
    mk0 = inp["mask0"].reshape(cfg["B"], -1) if cfg.get("masks") else None
    mk1 = inp["mask1"].reshape(cfg["B"], -1) if cfg.get("masks") else None
    audit_index_mismatches(N(st["next_idx_c01"]), g["next_idx_c01"], dot_score_fn(inp["feat0"], inp["feat1"], mk0, mk1), "next_idx_c01")
    audit_index_mismatches(N(st["next_idx_c10"]), g["next_idx_c10"], dot_score_fn(inp["feat1"], inp["feat0"], mk1, mk0), "next_idx_c10")
    assert_close(N(st["next_conf_c01"]), g["next_conf_c01"], TOL, "next_conf_c01")
    if "conf_matrix" in g:
        assert_close(N(st["conf_matrix"]), g["conf_matrix"], TOL, "conf_matrix")
    got, want = match_set(N(st["b_ids"]), N(st["i_ids"]), N(st["j_ids"])), match_set(g["b_ids"], g["i_ids"], g["j_ids"])
    assert len(got ^ want) <= 1
    if got == want:
        assert_close(N(st["mkpts0_c"]), g["mkpts0_c"], 1e-6, "mkpts0_c")
        assert_close(N(st["mkpts1_c"]), g["mkpts1_c"], 1e-6, "mkpts1_c")
        assert_close(N(st["mconf"]), g["mconf"], TOL, "mconf")


@pytest.mark.parametrize("name", list(CASES["cascade_matching"]))
def test_cascade_matching_module(name):
    from casmtr_amd.matching.cascade_matching import CascadeMatching
    inp = make_inputs("cascade_matching", name)
    cfg = CASES["cascade_matching"][name]
    g = load_golden("cascade_matching", name)
    hc, wc = cfg["coarse_hw"]
    h, w = 2 * hc, 2 * wc
    mcfg = {"thr": 0.2, "test_thr": cfg.get("test_thr", 0.2), "pre_thr": [cfg.get("pre_thr", 0.2)],
            "border_rm": cfg.get("border_rm", 2), "double_check": cfg.get("double_check", True),
            "train_pad_num_gt_min": 200, "match_type": "softmax", "dsmax_temperature": 1.0}
    post = {"method": "maxpool_nms", "window_size": 5} if cfg.get("nms", True) else {"method": None}
    post = cfg.get("post", post)   # 'local_window_nms' (§8 f.4)
    cm = CascadeMatching(mcfg, {"propagation": "window", "dilated": 1, "post_config": post}, stage="4c", div_mode="cpu").eval()
    data = {"hw0_i": (h * 4, w * 4), "hw1_i": (h * 4, w * 4), "hw0_8c": (hc, wc), "hw1_8c": (hc, wc), "hw0_4c": (h, w),
            "hw1_4c": (h, w), "stage_8c": {"next_conf_c01": T(inp["pre_conf"])}}
    m0 = m1 = None
    if cfg.get("masks"):
        data["mask_4c0"], data["mask_4c1"] = T(inp["mask0"]).bool(), T(inp["mask1"]).bool()
        m0, m1 = data["mask_4c0"].flatten(-2), data["mask_4c1"].flatten(-2)
    with torch.no_grad():
        cm(T(inp["feat0"]), T(inp["feat1"]), T(g["idx_c01"].astype(np.int64)), T(g["idx_c10"].astype(np.int64)), data,
           mask_c0=m0, mask_c1=m1, level="4c", pre_level="8c")
    st = data["stage_4c"]
    assert_close(N(st["conf_matrix"]), g["conf_matrix"], TOL, "conf_matrix01")
    assert_close(N(st["next_conf_c01"]), g["next_conf_c01"], TOL, "next_conf_c01")
    mq = inp["mask0"].reshape(cfg["B"], -1) if cfg.get("masks") else None
    mk = inp["mask1"].reshape(cfg["B"], -1) if cfg.get("masks") else None
    audit_index_mismatches(N(st["next_idx_c01"]), g["next_idx_c01"], dot_score_fn(inp["feat0"], inp["feat1"], mq, mk), "cascade next_idx_c01")
    audit_index_mismatches(N(st["next_idx_c10"]), g["next_idx_c10"], dot_score_fn(inp["feat1"], inp["feat0"], mk, mq), "cascade next_idx_c10")
    got, want = match_set(N(st["b_ids"]), N(st["i_ids"]), N(st["j_ids"])), match_set(g["b_ids"], g["i_ids"], g["j_ids"])
    # selection thresholds act on softmax values that differ by ulps between CPU and GPU expf -> borderline only
    assert len(got ^ want) <= max(1, len(want) // 100), (len(got), len(want))
    assert "m_bids" in data


def test_compat_reference_names_on_gpu():
    import casmtr_amd.compat as compat
    compat.install()
    import fast_score_computation
    import score_computation_cuda
    import value_aggregation_cuda
    inp = make_inputs("ops", "k64_h8")
    s = score_computation_cuda.score_forward(T(inp["q"]), T(inp["key"]), T(inp["idx"]))
    assert isinstance(s, list) and np.array_equal(N(s[0]), oracle.qta_score_fwd(inp["q"], inp["key"], inp["idx"]))
    w = fast_score_computation.score_forward(T(inp["wq"]), T(inp["wkey"]), T(inp["widx"]))
    assert np.array_equal(N(w[0]), oracle.window_score_fwd(inp["wq"], inp["wkey"], inp["widx"]))
    A = torch.rand(s[0].shape, device=DEV)
    B, N1, _, K, H = A.shape
    out = torch.zeros((B, N1 * 4, H, 32), device=DEV)
    idx5 = T(np.repeat(inp["idx"][:, :, None], 4, axis=2).reshape(B, N1 * 4, K, H))
    assert value_aggregation_cuda.value_aggregation_forward(A.view(B, N1 * 4, K, H), T(inp["value"]), idx5, out) is None
    assert out.abs().sum() > 0


def _valid_hw(m0, m1):
    return np.stack([m0.sum(1).max(-1), m0.sum(2).max(-1), m1.sum(1).max(-1), m1.sum(2).max(-1)], 1).astype(np.int32)


@pytest.mark.parametrize("masked", [False, True], ids=["configs1_unmasked", "configs2_padding_masks"])
def test_full_size_chain_vs_oracle(masked):
    """BASELINE configs[1] / configs[2] shapes (CasMTR-4c, 832x832, batch of 8; configs[2]: MegaDepth-style padding masks):
    every index output of the chain bit-matches the oracle, for the FIRST and the LAST pair of the batch."""
    from casmtr_amd import ops
    from casmtr_amd.pipeline import HotPath, HotPathConfig, make_synthetic_inputs
    B = 8
    cfg = HotPathConfig(masked=masked, implicit_windows=False)   # explicit int64 windows: data['stage_4c']['idx_c01'] is checked too
    model = HotPath(cfg).to(DEV)
    inp = make_synthetic_inputs(cfg, B, DEV, seed=7)
    with torch.no_grad():
        model.qta.weight.copy_(inp["weight"])
        out = model(inp)
        out_imp = HotPath(HotPathConfig(masked=masked, implicit_windows=True)).to(DEV)
        out_imp.qta.weight.copy_(inp["weight"])
        out_imp = out_imp(inp)
    torch.cuda.synchronize()
    d = out["data"]
    st8, st4 = d["stage_8c"], d["stage_4c"]
    # the implicit-window data flow (topk_pos instead of the int64 index tensor) gives the same answers, bit for bit
    for k in ("next_idx_c01", "next_idx_c10", "next_conf_c01", "conf_matrix", "b_ids", "i_ids", "j_ids", "mconf"):
        assert torch.equal(out_imp["data"]["stage_4c"][k], st4[k]), f"implicit windows: stage_4c[{k}] differs"
    assert torch.equal(out_imp["data"]["stage_4c"]["idx_c01"].materialize(), st4["idx_c01"])
    # QTAttB per-level indices at 26x26 / 52x52 / 104x104 for the whole batch (layer 1 = first cross layer, direction 0 -> 1)
    toks = ops.nchw_to_tokens_multi([x.contiguous() for lv in (2, 1, 0) for x in (inp["cq0"][1][lv], inp["ck1"][1][lv], inp["cv1"][1][lv])])
    levels_gpu, prev = [], None
    for i in range(3):
        q, k, v = toks[3 * i:3 * i + 3]
        hw = tuple(inp["cq0"][1][2 - i].shape[2:])
        o = (ops.qta_coarse_level(q, k, v, cfg.coarse_heads, cfg.coarse_topks[0]) if i == 0 else
             ops.qta_fine_level(q, k, v, prev, hw, hw, cfg.coarse_heads, cfg.coarse_topks[i]))
        prev = o["topk_idx"]
        levels_gpu.append(o)
    # the same call through the module's default path: finer levels on quad-major operands (csrc/fine_quad.hip), top-k lists
    # as compact int32 tables between the levels; the reference's int64 tensors materialised for this comparison only
    lv_nchw = [(inp["cq0"][1][lv], inp["ck1"][1][lv], inp["cv1"][1][lv]) for lv in (2, 1, 0)]
    hws = [tuple(t[0].shape[2:]) for t in lv_nchw]
    assert model.qta._quad_major_ok(hws, hws), "the headline shapes must run the quad-major kernel"
    with torch.no_grad():
        final_qm = model.qta._fused_levels_quad(lv_nchw, hws, hws, want_topk=True)
    levels_qm = model.qta._last_levels
    assert torch.equal(final_qm, out["messages"][2]), "materialising the top-k tensors must not change the message"
    for e in (0, B - 1):
        n = lambda t: N(t[e:e + 1])
        fo, lv = oracle.qtattb_forward([n(x) for x in inp["cq0"][1]], [n(x) for x in inp["ck1"][1]], [n(x) for x in inp["cv1"][1]],
                                       N(inp["weight"]), cfg.coarse_heads, cfg.coarse_topks)
        for i, side in enumerate((26, 52, 104)):
            assert np.array_equal(n(levels_gpu[i]["topk_idx"]), lv[i]["topk_idx"]), f"pair {e}: top-k indices at {side}x{side}"
            assert_close(n(levels_gpu[i]["topk_score"]), lv[i]["topk_score"], TOL, f"pair {e}: top-k scores at {side}x{side}")
            if i < 2:   # (the module skips the finest level's top-k, which the reference computes and discards)
                assert np.array_equal(n(levels_qm[i]["topk_idx"]), lv[i]["topk_idx"]), f"pair {e}: quad-major path, top-k at {side}x{side}"
                assert np.array_equal(n(levels_qm[i]["topk_tab"]), lv[i]["topk_idx"].transpose(0, 3, 1, 2)), f"pair {e}: int32 table at {side}x{side}"
                assert_close(n(levels_qm[i]["topk_score"]), lv[i]["topk_score"], TOL, f"pair {e}: quad-major top-k scores at {side}x{side}")
        assert_close(n(out["messages"][2]), fo, TOL, f"pair {e}: QTAttB cross message (104x104)")
        # coarse matching
        mk = (lambda lvl, im: n(inp[f"mask_{lvl}{im}"]).reshape(1, -1)) if masked else (lambda lvl, im: None)
        vh = (lambda lvl: _valid_hw(n(inp[f"mask_{lvl}0"]), n(inp[f"mask_{lvl}1"]))) if masked else (lambda lvl: None)
        o8 = oracle.dual_softmax(n(inp["feat_8c0"]), n(inp["feat_8c1"]), cfg.hw8, cfg.hw8, cfg.coarse_temperature, cfg.coarse_thr,
                                 cfg.coarse_border_rm, mask0=mk("8c", 0), mask1=mk("8c", 1), valid_hw=vh("8c"), recip=True)
        assert np.array_equal(n(st8["next_idx_c01"]), o8["next_idx_c01"]), f"pair {e}: coarse row argmax (10816 x 10816)"
        assert np.array_equal(n(st8["next_idx_c10"]), o8["next_idx_c10"]), f"pair {e}: coarse column argmax"
        assert_close(n(st8["next_conf_c01"]), o8["next_conf_c01"], TOL, "coarse next_conf")
        sel = N(st8["b_ids"]) == e
        got = match_set(0 * N(st8["i_ids"])[sel], N(st8["i_ids"])[sel], N(st8["j_ids"])[sel])
        want = match_set(o8["b_ids"], o8["i_ids"], o8["j_ids"])
        assert len(got ^ want) <= 1 and len(want) > 1000, "coarse match list (only a conf == thr borderline entry may differ)"
        # cascade attention + matching
        tp01 = oracle.window_warp_idx(o8["next_idx_c01"], *cfg.hw8, cfg.window_size)
        tp10 = oracle.window_warp_idx(o8["next_idx_c10"], *cfg.hw8, cfg.window_size)
        tok = lambda x: np.ascontiguousarray(n(x).transpose(0, 2, 3, 1).reshape(1, -1, cfg.cascade_dim))
        mo, i01 = oracle.cascade_attn(tok(inp["4cq0"][0]), tok(inp["4ck1"][0]), tok(inp["4cv1"][0]), tp01, cfg.hw4, cfg.hw4, cfg.cascade_heads)
        _, i10 = oracle.cascade_attn(tok(inp["4cq1"][0]), tok(inp["4ck0"][0]), tok(inp["4cv0"][0]), tp10, cfg.hw4, cfg.hw4, cfg.cascade_heads)
        assert np.array_equal(n(st4["idx_c01"]), i01), f"pair {e}: upsampled window indices"
        assert_close(n(out["messages"][12]), mo, TOL, f"pair {e}: CascadeQTAttB message (208x208)")
        m01 = oracle.window_match(n(inp["feat_4c0"]), n(inp["feat_4c1"]), i01, cfg.cascade_temperature, mk("4c", 0), mk("4c", 1), recip=True)
        m10 = oracle.window_match(n(inp["feat_4c1"]), n(inp["feat_4c0"]), i10, cfg.cascade_temperature, mk("4c", 1), mk("4c", 0), recip=True,
                                  want_conf=False)
        assert np.array_equal(n(st4["next_idx_c01"]), m01["next_idx"]), f"pair {e}: cascade argmax 0->1 (43264 x 100)"
        assert np.array_equal(n(st4["next_idx_c10"]), m10["next_idx"]), f"pair {e}: cascade argmax 1->0"
        assert_close(n(st4["conf_matrix"]), m01["conf_matrix"], TOL, "cascade conf_matrix")
        so = oracle.nms_select(N(st4["next_conf_c01"][e:e + 1]), m01["next_idx"], m10["next_idx"], cfg.hw4, cfg.hw4, cfg.nms_window,
                               cfg.cascade_test_thr, [(N(st8["next_conf_c01"][e:e + 1]), cfg.hw8, cfg.cascade_pre_thr)],
                               cfg.cascade_border_rm, valid_hw=vh("4c"))
        sel = N(st4["b_ids"]) == e
        assert np.array_equal(N(st4["i_ids"])[sel], so["i_ids"]) and np.array_equal(N(st4["j_ids"])[sel], so["j_ids"]), \
            f"pair {e}: final match list (NMS, thresholds, borders{', padding' if masked else ''}, double check)"
        assert sel.sum() > 100
    # size-independent properties: every kept match survives its own definition
    i_ids, j_ids, b_ids = N(st4["i_ids"]), N(st4["j_ids"]), N(st4["b_ids"])
    ni01, ni10, nc01 = N(st4["next_idx_c01"]), N(st4["next_idx_c10"]), N(st4["next_conf_c01"])
    assert np.array_equal(ni01[b_ids, i_ids], j_ids)
    assert np.array_equal(ni10[b_ids, j_ids], i_ids), "double check"
    assert (nc01[b_ids, i_ids] > cfg.cascade_test_thr).all()
    assert (np.diff(b_ids * 10**6 + i_ids) > 0).all(), "(b,i) ordering"
    if masked:   # no match may start or end in the padding
        m0, m1 = N(inp["mask_4c0"]).reshape(B, -1), N(inp["mask_4c1"]).reshape(B, -1)
        assert m0[b_ids, i_ids].all() and m1[b_ids, j_ids].all()


def _rand(shape, seed):
    return np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)


def test_2c_level_shapes_vs_oracle():
    """BASELINE configs[3] (CasMTR-2c, 1/2-res cascade): H = 2, C = 64, 416x416 query grid, K = 100."""
    from casmtr_amd import ops
    hc, wc, H, C = 208, 208, 2, 64
    h, w = 2 * hc, 2 * wc
    r = np.random.default_rng(11)
    coarse_idx = r.integers(0, hc * wc, (1, hc * wc), dtype=np.int64)
    tp = ops.window_warp_idx(T(coarse_idx), hc, wc, 5)
    assert np.array_equal(N(tp), oracle.window_warp_idx(coarse_idx, hc, wc, 5))
    q, k, v = (_rand((1, h * w, C), s) for s in (1, 2, 3))
    msg, up = ops.cascade_attn(T(q), T(k), T(v), tp, (h, w), (h, w), H)
    mo, uo = oracle.cascade_attn(q, k, v, N(tp), (h, w), (h, w), H)
    assert np.array_equal(N(up), uo)
    assert_close(N(msg), mo, TOL, "2c cascade message (173056 tokens)")
    f0, f1 = 3.0 * _rand((1, h * w, C), 4), 3.0 * _rand((1, h * w, C), 5)
    d = ops.window_match(T(f0), T(f1), up, 1.0, recip=True, hw=(h, w))
    o = oracle.window_match(f0, f1, uo, 1.0, recip=True)
    assert np.array_equal(N(d["next_idx"]), o["next_idx"]), "2c window argmax (173056 x 100) must be bit-exact"
    assert_close(N(d["conf_matrix"]), o["conf_matrix"], TOL, "2c conf")
    # two previous stages (8c and 4c) feed the 2c selection (pre_thr = [[0.2],[0.2,0.2]], NMS on 2c only)
    pre8, pre4 = r.random((1, 104 * 104), dtype=np.float32), r.random((1, hc * wc), dtype=np.float32)
    sel = ops.nms_select(d["next_conf"], d["next_idx"], d["next_idx"], (h, w), (h, w), nms_window=5, test_thr=0.05,
                         pre=[(T(pre8), (104, 104), 0.2), (T(pre4), (hc, wc), 0.2)], border_rm=2, double_check=False)
    so = oracle.nms_select(N(d["next_conf"]), N(d["next_idx"]), N(d["next_idx"]), (h, w), (h, w), nms_window=5, test_thr=0.05,
                           pre=[(pre8, (104, 104), 0.2), (pre4, (hc, wc), 0.2)], border_rm=2, double_check=False)
    n = int(sel["n"].item())
    assert n == len(so["b_ids"]) and n > 50
    assert np.array_equal(N(sel["i_ids"][:n]), so["i_ids"]) and np.array_equal(N(sel["j_ids"][:n]), so["j_ids"])


def test_indoor_shapes_vs_oracle():
    """BASELINE configs[4] shapes (ScanNet indoor 640x480): 80x60 coarse grid, topks [32,16,16], rel_pos in the cascade."""
    from casmtr_amd import ops
    from casmtr_amd.modules.quadtree_attention import CascadeQTAttB, QTAttB
    H, D = 8, 32
    hh, ww = 60, 80

    def pyr(seed):
        x = _rand((1, H * D, hh, ww), seed)
        out = [x]
        for _ in range(2):
            b, c, a, bb = x.shape
            x = np.ascontiguousarray(x.reshape(b, c, a // 2, 2, bb // 2, 2).mean(axis=(3, 5), dtype=np.float32))
            out.append(x)
        return out

    qs, ks, vs = pyr(21), pyr(22), pyr(23)
    wt = _rand((3,), 24)
    m = QTAttB(H, D, scale=3, topks=[32, 16, 16]).to(DEV).eval()
    with torch.no_grad():
        m.weight.copy_(T(wt))
        out = m([T(x) for x in qs], [T(x) for x in ks], [T(x) for x in vs])
    fo, lv = oracle.qtattb_forward(qs, ks, vs, wt, H, [32, 16, 16])
    assert_close(N(out), fo, TOL, "indoor QTAttB message (80x60)")
    # per-level indices through the ops (the module does not expose them)
    tok = lambda x: np.ascontiguousarray(x.transpose(0, 2, 3, 1).reshape(1, -1, H * D))
    c = ops.qta_coarse_level(T(tok(qs[2])), T(tok(ks[2])), T(tok(vs[2])), H, 32)
    assert np.array_equal(N(c["topk_idx"]), lv[0]["topk_idx"])
    f1 = ops.qta_fine_level(T(tok(qs[1])), T(tok(ks[1])), T(tok(vs[1])), c["topk_idx"], (30, 40), (30, 40), H, 16)
    assert np.array_equal(N(f1["topk_idx"]), lv[1]["topk_idx"]), "indoor level-1 top-16 of 128"
    # cascade with relative position bias: [B, nhead, H0*W0, 4*ww]
    hc, wc, Hc = 60, 80, 4
    h, w = 2 * hc, 2 * wc
    r = np.random.default_rng(31)
    tp = oracle.window_warp_idx(r.integers(0, hc * wc, (1, hc * wc), dtype=np.int64), hc, wc, 5)
    q, k, v = (_rand((1, 128, h, w), s) for s in (32, 33, 34))
    rel = _rand((1, Hc, h * w, 100), 35)
    cm = CascadeQTAttB(Hc, 32, dilated=1).to(DEV)
    with torch.no_grad():
        msg, up = cm(T(q), T(k), T(v), T(tp), T(rel))
    tk = lambda x: np.ascontiguousarray(x.transpose(0, 2, 3, 1).reshape(1, h * w, 128))
    mo, uo = oracle.cascade_attn(tk(q), tk(k), tk(v), tp, (h, w), (h, w), Hc, rel_pos=rel)
    assert np.array_equal(N(up), uo)
    assert_close(N(msg), mo, TOL, "indoor cascade message with rel_pos (160x120)")


def _audit_match_list_diff(got, want, conf_g, conf_o, hw, st, pre_g, what, eps=2e-5):
    """got / want: sets of (i, j) of one pair at one stage (GPU chain vs the independent oracle chain).  The index tensors feeding
    the selection are asserted EQUAL by the caller, so the two lists can only differ where a float comparison sits within the
    softmax tolerance of its threshold: conf vs test_thr, a previous stage's conf vs pre_thr, or two confidences of one NMS
    window within eps of each other (the reference's max-pool winner then depends on the last bits of expf).  Every differing
    entry must show one of those causes."""
    h, w = hw
    cg, co = conf_g.reshape(h, w), conf_o.reshape(h, w)
    for i, j in sorted(got ^ want):
        y, x = divmod(i, w)
        why = []
        if min(abs(cg[y, x] - st.test_thr), abs(co[y, x] - st.test_thr)) <= eps:
            why.append("conf ~ test_thr")
        for (pc, (ph, pw)), thr in zip(pre_g, st.pre_thr):
            v = pc.reshape(ph, pw)[y * ph // h, x * pw // w]
            if abs(v - thr) <= eps:
                why.append("previous-stage conf ~ pre_thr")
        if st.nms_window:
            r = st.nms_window // 2
            win = co[max(y - r, 0):y + r + 1, max(x - r, 0):x + r + 1]
            if (np.abs(win - co[y, x]) <= eps).sum() > 1 and co[y, x] >= win.max() - eps:
                why.append("NMS window near-tie")
        assert why, f"{what}: match ({i}, {j}) differs between the GPU chain and the oracle chain without a borderline cause"
    return len(got ^ want)


@pytest.mark.parametrize("which", ["2c", "indoor"])
def test_named_config_chain_vs_oracle(which):
    """BASELINE configs[3] (CasMTR-2c: 4c stage without NMS and border_rm 1, then the 2c stage with pre_level ['8c','4c'], NMS,
    cascade_model_stage4.py:150-195) and configs[4] shapes (indoor 640x480: topks [32,16,16], 8 coarse layers, rel_pos, no NMS) as
    whole chains with the SHIPPED thresholds, B = 2, first and last pair: every index output equals the oracle chain's, values
    within 1e-4, match lists exact given the GPU's own confidences and audited against the independent oracle chain."""
    from casmtr_amd.pipeline import HotPath, HotPathConfig, make_synthetic_inputs
    from oracle.chain import run_chain
    B = 2
    cfg = HotPathConfig.named(which)                                  # implicit windows: the shipped data flow
    cfg_e = HotPathConfig.named(which, implicit_windows=False)        # the reference's int64 window tensors, for the index check
    inp = make_synthetic_inputs(cfg, B, DEV, seed=23)
    outs = []
    with torch.no_grad():
        for c in (cfg, cfg_e):
            m = HotPath(c).to(DEV)
            m.qta.weight.copy_(inp["weight"])
            outs.append(m(inp))
        torch.cuda.synchronize()
        out, out_e = outs
        # per-level top-k of the first cross call through the module's own (quad-major) path
        lv_nchw = [(inp["cq0"][1][lv], inp["ck1"][1][lv], inp["cv1"][1][lv]) for lv in (2, 1, 0)]
        hws = [tuple(t[0].shape[2:]) for t in lv_nchw]
        assert m.qta._quad_major_ok(hws, hws)
        m.qta._fused_levels_quad(lv_nchw, hws, hws, want_topk=True)
        levels_qm = m.qta._last_levels
    for lvl in [st.level for st in cfg.stages]:
        for k in ("next_idx_c01", "next_idx_c10", "next_conf_c01", "b_ids", "i_ids", "j_ids", "mconf"):
            assert torch.equal(out["data"][f"stage_{lvl}"][k], out_e["data"][f"stage_{lvl}"][k]), f"implicit vs explicit windows: stage_{lvl}[{k}]"
    n_calls = 2 * cfg.coarse_layers
    for e in (0, B - 1):
        n = lambda t: N(t[e:e + 1])
        o = run_chain(cfg, inp, pair=e, qta_calls=(0, 2, n_calls - 1))
        for call, (fo, lv) in o["qta"].items():
            assert_close(n(out["messages"][call]), fo, TOL, f"pair {e}: QTAttB call {call}")
        for i in range(2):
            assert np.array_equal(n(levels_qm[i]["topk_idx"]), o["qta"][2][1][i]["topk_idx"]), f"pair {e}: QTAttB top-k at level {i}"
        st8, d8 = out["data"]["stage_8c"], o["d8"]
        assert np.array_equal(n(st8["next_idx_c01"]), d8["next_idx_c01"]) and np.array_equal(n(st8["next_idx_c10"]), d8["next_idx_c10"])
        assert_close(n(st8["next_conf_c01"]), d8["next_conf_c01"], TOL, "coarse next_conf")
        pre_g = [(n(st8["next_conf_c01"]), cfg.hw8)]
        mi = n_calls
        for st in cfg.stages:
            lvl, hw = st.level, cfg.hw(st.div)
            g, ge, so = out["data"][f"stage_{lvl}"], out_e["data"][f"stage_{lvl}"], o["stages"][lvl]
            assert np.array_equal(n(ge["idx_c01"]), so["i01"]) and np.array_equal(n(ge["idx_c10"]), so["i10"]), f"pair {e}: {lvl} window indices"
            for j, mo in enumerate(so["msgs"]):
                assert_close(n(out["messages"][mi + j]).reshape(mo.shape), mo, TOL, f"pair {e}: {lvl} CascadeQTAttB message {j}")
            mi += len(so["msgs"])
            assert np.array_equal(n(g["next_idx_c01"]), so["m01"]["next_idx"]), f"pair {e}: {lvl} window argmax 0->1"
            assert np.array_equal(n(g["next_idx_c10"]), so["m10"]["next_idx"]), f"pair {e}: {lvl} window argmax 1->0"
            assert_close(n(g["conf_matrix"]), so["m01"]["conf_matrix"], TOL, f"{lvl} conf_matrix")
            # (1) the selection itself, given the GPU's own confidences: exact
            sel_g = oracle.nms_select(n(g["next_conf_c01"]), so["m01"]["next_idx"], so["m10"]["next_idx"], hw, hw, st.nms_window, st.test_thr,
                                      [(pc, phw, thr) for (pc, phw), thr in zip(pre_g, st.pre_thr)], st.border_rm)
            mine = N(g["b_ids"]) == e
            assert np.array_equal(N(g["i_ids"])[mine], sel_g["i_ids"]) and np.array_equal(N(g["j_ids"])[mine], sel_g["j_ids"]), \
                f"pair {e}: {lvl} match list (thresholds {st.test_thr} / {st.pre_thr}, border_rm {st.border_rm}, NMS {st.nms_window}, double check)"
            assert mine.sum() > 50
            # (2) against the independent oracle chain (its own softmax values): differences only at borderline comparisons
            got = set(zip(N(g["i_ids"])[mine].tolist(), N(g["j_ids"])[mine].tolist()))
            want = set(zip(so["sel"]["i_ids"].tolist(), so["sel"]["j_ids"].tolist()))
            nd = _audit_match_list_diff(got, want, n(g["next_conf_c01"]), so["m01"]["next_conf"], hw, st, pre_g, f"pair {e}, stage {lvl}")
            assert nd <= max(2, len(want) // 500)
            pre_g.append((n(g["next_conf_c01"]), hw))


@pytest.mark.parametrize("name", list(CASES["qtatt_variants"]))
def test_qtatt_variants_vs_reference(name):
    """QTAttA and QTAttGuided (no shipped config selects them) against the reference python."""
    from casmtr_amd.modules.quadtree_attention import QTAttA, QTAttGuided
    inp = make_inputs("qtatt_variants", name)
    cfg = CASES["qtatt_variants"][name]
    g = load_golden("qtatt_variants", name)
    qs, ks, vs = ([T(x) for x in inp[n]] for n in ("queries", "keys", "values"))
    with torch.no_grad():
        if cfg["kind"] == "A":
            out = QTAttA(cfg["nhead"], cfg["D"], topks=cfg["topks"]).to(DEV)(qs, ks, vs)
        else:
            m = QTAttGuided(cfg["nhead"], cfg["D"], scale=len(cfg["topks"]), topks=cfg["topks"]).to(DEV)
            m.weight.copy_(T(inp["weight"]))
            out = m(qs, ks, vs, topk_pos=T(inp["topk_pos"]))
    assert out.shape == g["final"].shape
    assert_close(N(out), g["final"], TOL, f"{cfg['kind']} final message vs reference python")


# ---------------------------------------------------------------------------------------------------------------------
# size-independent properties at BASELINE's full size (832x832, batch of 8): no oracle needed, so the whole batch is covered
def _snapshot(out):
    """every tensor the step produces, flattened into {name: tensor}"""
    snap = {f"message{i}": m for i, m in enumerate(out["messages"])}
    for lvl in ("8c", "4c"):
        st = out["data"][f"stage_{lvl}"]
        for k in ("next_idx_c01", "next_idx_c10", "next_conf_c01", "next_conf_c10", "b_ids", "i_ids", "j_ids", "mconf"):
            if torch.is_tensor(st.get(k)):
                snap[f"{lvl}.{k}"] = st[k]
    return snap


def _run_full(cfg_kw=None, B=8, seed=11, env=None, monkeypatch=None, select=None):
    from casmtr_amd.pipeline import HotPath, HotPathConfig, make_synthetic_inputs
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    cfg = HotPathConfig(**(cfg_kw or {}))
    model = HotPath(cfg).to(DEV)
    inp = make_synthetic_inputs(cfg, 8, DEV, seed=seed)
    if select is not None:   # a sub-batch of the same pairs
        inp = {k: (_take(v, select)) for k, v in inp.items()}
    with torch.no_grad():
        model.qta.weight.copy_(inp["weight"])
        out = model(inp)
    torch.cuda.synchronize()
    return _snapshot(out)


def _take(v, sel):
    if torch.is_tensor(v):
        return v[sel].contiguous() if v.dim() > 1 and v.shape[0] == 8 else v
    if isinstance(v, (list, tuple)):
        return [_take(x, sel) for x in v]
    return v


def test_full_size_run_to_run_determinism():
    """the persistent kernels hand out work dynamically (cursor per wave, XCD-chunked lists): two runs on the same inputs must
    still agree bit for bit in every output, values included"""
    a, b = _run_full(), _run_full()
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), f"{k} differs between two identical runs"


def test_full_size_pairs_are_independent_of_their_batch():
    """a pair's results do not depend on which other pairs share its launch: pairs 5 and 2 run alone (in that order) give the
    rows they get inside the batch of 8 -- every message, every argmax, every confidence, bit for bit"""
    full = _run_full()
    sel = torch.tensor([5, 2], device=DEV)
    sub = _run_full(select=sel)
    for k, v in sub.items():
        if k.endswith(("b_ids", "i_ids", "j_ids", "mconf")):
            continue   # match lists are compacted over the batch; compared below per pair
        assert torch.equal(v, full[k][sel]), f"{k}: pair results depend on the batch"
    for lvl in ("8c", "4c"):
        for new_b, old_b in enumerate((5, 2)):
            m_sub = sub[f"{lvl}.b_ids"] == new_b
            m_full = full[f"{lvl}.b_ids"] == old_b
            for k in ("i_ids", "j_ids", "mconf"):
                assert torch.equal(sub[f"{lvl}.{k}"][m_sub], full[f"{lvl}.{k}"][m_full]), f"{lvl}.{k} of pair {old_b}"


@pytest.mark.parametrize("env", [{"CASMTR_FINE_KERNEL": "dma"}, {"CASMTR_FINE_KERNEL": "quad"}, {"CASMTR_CASCADE_KERNEL": "quad"},
                                 {"CASMTR_WINDOW_KERNEL": "quad"}, {"CASMTR_COARSE_KERNEL": "three"}, {"CASMTR_DS_GEMM": "exact"}],
                         ids=lambda e: "-".join(f"{k[7:].lower()}={v}" for k, v in e.items()))   # ds_gemm=exact: the all-fp32 dual-softmax GEMM against the f16 split
def test_full_size_kernel_variants_agree(monkeypatch, env):
    """every alternative kernel behind the CASMTR_*_KERNEL selectors reproduces the default kernels' indices exactly and their
    values within the softmax tolerance, on the whole full-size batch"""
    ref = _run_full()
    alt = _run_full(env=env, monkeypatch=monkeypatch)
    for k in ref:
        if ref[k].dtype in (torch.int64, torch.int32, torch.bool):
            assert torch.equal(ref[k], alt[k]), f"{k}: index output differs under {env}"
        else:
            assert ref[k].shape == alt[k].shape and float((ref[k] - alt[k]).abs().max()) <= TOL, f"{k} under {env}"


def test_hip_graph_replay_equals_eager():
    """casmtr_amd.graph.GraphedHotPath: a whole step captured into a HIP graph and replayed 40 times (lists read one step behind)
    returns the eager step's match lists bit for bit -- guards the capture-safety of every launch on the path (no hidden
    allocation, synchronisation or memset node: the dual-softmax workspace is cleared by a kernel for that reason)"""
    from casmtr_amd.graph import GraphedHotPath
    from casmtr_amd.pipeline import HotPath, HotPathConfig, make_synthetic_inputs
    cfg = HotPathConfig(name="small", image_hw=(256, 320), coarse_layers=2)
    model = HotPath(cfg).to(DEV)
    inp = make_synthetic_inputs(cfg, 2, DEV, seed=7)
    with torch.no_grad():
        model.qta.weight.copy_(inp["weight"])
    ref = model(inp)
    ref = {k: ref[k].clone() for k in ("m_bids", "mkpts0", "mkpts1", "mconf")}
    gs = GraphedHotPath(model, inp)
    pend = None
    for _ in range(40):
        new = gs.enqueue()
        if pend is not None:
            o = gs.finalize(pend)
            for k in ref:
                assert torch.equal(o[k], ref[k]), k
        pend = new
    o = gs.finalize(pend)
    assert all(torch.equal(o[k], ref[k]) for k in ref)
    assert int(ref["m_bids"].numel()) > 0


def test_run_ahead_side_stream_finalize_equals_eager():
    """casmtr_amd.pipeline.RunAhead (bench.py's timed loop): step k is finalised on a side stream while step k+1 is already
    enqueued; with inputs that alternate between two seeds every returned list must equal the eager step's on the same input"""
    from casmtr_amd.pipeline import HotPath, HotPathConfig, RunAhead, make_synthetic_inputs
    cfg = HotPathConfig(name="small", image_hw=(256, 320), coarse_layers=2)
    model = HotPath(cfg).to(DEV)
    inps = [make_synthetic_inputs(cfg, 2, DEV, seed=s) for s in (7, 8)]
    with torch.no_grad():
        model.qta.weight.copy_(inps[0]["weight"])
    keys = ("m_bids", "mkpts0", "mkpts1", "mconf")
    refs = []
    for inp in inps:
        r = model(inp)
        refs.append(({k: r[k].clone() for k in keys},
                     {k: r["data"]["stage_8c"][k].clone() for k in ("b_ids", "i_ids", "j_ids", "mconf", "m_bids", "mkpts0_c")}))
    assert not torch.equal(refs[0][0]["mconf"], refs[1][0]["mconf"])
    ra = RunAhead(model)
    outs = [ra.submit(inps[i & 1]) for i in range(24)] + [ra.drain()]
    assert outs[0] is None and ra.drain() is None
    for i, o in enumerate(outs[1:]):
        ref, ref8 = refs[i & 1]
        for k in keys:
            assert torch.equal(o[k], ref[k]), (i, k)
        for k in ref8:
            assert torch.equal(o["data"]["stage_8c"][k], ref8[k]), (i, k)


def test_run_ahead_results_dropped_behind_a_slow_consumer():
    """a caller may drop a finalised result right after enqueueing its launch-stream consumer
    (`consume(ra.submit(x))`, MatchGatherer.flush).  The lists live in the side stream's pool and are not record_stream-ed, so the
    NEXT finalize must not start before that consumer has run: RunAhead makes it wait for a launch-stream event recorded at the
    start of the submit that runs it.  Here every consumer sits behind ~10 ms of unrelated launch-stream work and the result is
    dropped at once; with the missing wait the following finalize rewrote the lists first."""
    from casmtr_amd.pipeline import HotPath, HotPathConfig, RunAhead, make_synthetic_inputs
    cfg = HotPathConfig(name="small", image_hw=(256, 320), coarse_layers=2)
    model = HotPath(cfg).to(DEV)
    inps = [make_synthetic_inputs(cfg, 2, DEV, seed=s) for s in (7, 8)]
    with torch.no_grad():
        model.qta.weight.copy_(inps[0]["weight"])
    keys = ("mkpts0", "mkpts1", "mconf")
    want = []
    for inp in inps:
        r = model(inp)
        want.append(torch.stack([r[k].double().sum() for k in keys]))
    assert not torch.equal(want[0], want[1])
    big = torch.randn(4096, 4096, device=DEV) / 64
    ra = RunAhead(model)
    sums = []

    def consume(res):
        if res is None:
            return
        x = big
        for _ in range(6):   # ~10 ms of launch-stream work in front of the consumer
            x = x @ big
        s = torch.stack([res[k].double().sum() for k in keys])   # same stream: runs behind the matmuls
        sums.append(s)       # the result itself goes out of scope here

    for i in range(16):
        consume(ra.submit(inps[i & 1]))
    consume(ra.drain())
    torch.cuda.synchronize()
    assert len(sums) == 16
    for i, s in enumerate(sums):
        assert torch.equal(s, want[i & 1]), (i, s, want[i & 1])
