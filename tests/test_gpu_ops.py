"""GPU parity: every C-ABI entry point of libcasmtr_hip.so against the CPU oracle on the same seeded inputs.

Bar: indices bit-exact (int64 compare), fp32 dot-product outputs bit-exact (same fmaf chain), softmax-derived values
within 1e-4 (north_star) -- in practice ~1e-6.
"""
import numpy as np
import pytest
import torch

import oracle
from golden_inputs import CASES, make_inputs
from parity_utils import assert_close, audit_index_mismatches, dot_score_fn, load_golden, match_set, post_extra_mask

pytestmark = pytest.mark.gpu
SOFTMAX_TOL = 1e-4
DEV = "cuda:0"


def T(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    return t if dtype is None else t.to(dtype)


def N(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
    from casmtr_amd import ops as o
    return o


@pytest.mark.parametrize("name", list(CASES["ops"]))
def test_primitives_bit_exact(ops, name):
    inp = make_inputs("ops", name)
    s = ops.qta_score_fwd(T(inp["q"]), T(inp["key"]), T(inp["idx"]))
    so = oracle.qta_score_fwd(inp["q"], inp["key"], inp["idx"])
    assert np.array_equal(N(s), so), "qta_score_fwd must be bit-exact (same fmaf chain)"
    A = np.random.default_rng(0).random(so.shape, dtype=np.float32)
    B, N1, _, K, H = A.shape
    idx5 = np.repeat(inp["idx"][:, :, None], 4, axis=2).reshape(B, N1 * 4, K, H)
    out = torch.zeros((B, N1 * 4, H, inp["value"].shape[-1]), device=DEV)
    ops.qta_value_agg_fwd(T(A.reshape(B, N1 * 4, K, H)), T(inp["value"]), T(idx5), out)
    mo = oracle.qta_value_agg_fwd(A.reshape(B, N1 * 4, K, H), inp["value"], idx5)
    assert np.array_equal(N(out), mo), "qta_value_agg_fwd must be bit-exact (sequential-k fmaf chain)"
    ws = ops.window_score_fwd(T(inp["wq"]), T(inp["wkey"]), T(inp["widx"]))
    assert np.array_equal(N(ws), oracle.window_score_fwd(inp["wq"], inp["wkey"], inp["widx"]))
    # golden (reference python) within fp32 summation-order noise
    g = load_golden("ops", name)
    assert_close(N(s), g["score"], 5e-5, "score vs reference python")


@pytest.mark.parametrize("name", list(CASES["ops"]))
def test_primitives_backward(ops, name):
    inp = make_inputs("ops", name)
    r = np.random.default_rng(1)
    B, N1, _, H, D = inp["q"].shape
    K = inp["idx"].shape[2]
    g = r.standard_normal((B, N1, 4, K, H), dtype=np.float32)
    dq, dk = ops.qta_score_bwd(T(g), T(inp["q"]), T(inp["key"]), T(inp["idx"]))
    dqo, dko = oracle.qta_score_bwd(g, inp["q"], inp["key"], inp["idx"])
    assert_close(N(dq), dqo, 2e-4, "score dq")
    assert_close(N(dk), dko, 2e-4, "score dkey (atomics)")
    sc = r.random((B, N1 * 4, K, H), dtype=np.float32)
    idx5 = np.repeat(inp["idx"][:, :, None], 4, axis=2).reshape(B, N1 * 4, K, H)
    go = r.standard_normal((B, N1 * 4, H, D), dtype=np.float32)
    gs = torch.zeros((B, N1 * 4, K, H), device=DEV)
    gv = torch.zeros(inp["value"].shape, device=DEV)
    ops.qta_value_agg_bwd(T(go), T(sc), T(inp["value"]), T(idx5), gs, gv)
    gso, gvo = oracle.qta_value_agg_bwd(go, sc, inp["value"], idx5)
    assert_close(N(gs), gso, 2e-4, "agg grad_score")
    assert_close(N(gv), gvo, 2e-4, "agg grad_value (atomics)")
    gw = r.standard_normal(inp["widx"].shape, dtype=np.float32)
    dq, dk = ops.window_score_bwd(T(gw), T(inp["wq"]), T(inp["wkey"]), T(inp["widx"]))
    dqo, dko = oracle.window_score_bwd(gw, inp["wq"], inp["wkey"], inp["widx"])
    assert_close(N(dq), dqo, 5e-4, "window dq")
    assert_close(N(dk), dko, 5e-4, "window dkey (atomics)")


def test_nchw_to_tokens(ops):
    x = np.random.default_rng(2).standard_normal((2, 70, 9, 13), dtype=np.float32)
    out = ops.nchw_to_tokens(T(x))
    assert np.array_equal(N(out), x.transpose(0, 2, 3, 1).reshape(2, 9 * 13, 70))


def test_channels_last_is_zero_copy(ops):
    x = torch.randn(2, 64, 6, 10, device=DEV)
    xl = x.contiguous(memory_format=torch.channels_last)
    a, b = ops.nchw_to_tokens_multi([x, xl])
    assert torch.equal(a, b) and b.data_ptr() == xl.data_ptr() and a.data_ptr() != x.data_ptr()
    assert b.is_contiguous() and b.shape == (2, 60, 64)


def _tok(x):
    B, C, h, w = x.shape
    return np.ascontiguousarray(x.transpose(0, 2, 3, 1).reshape(B, h * w, C))


def _fine_level(ops, kernel, q, k, v, prev, hw0, hw1, H, topk, **kw):
    """kernel 'qm': the round-3 quad-major kernel (operands re-laid-out on the GPU, previous top-k as the compact int32 table);
    otherwise the token-major entry point with CASMTR_FINE_KERNEL = kernel"""
    if kernel != "qm":
        return ops.qta_fine_level(q, k, v, prev, hw0, hw1, H, topk, **kw)
    return ops.qta_fine_level_quad(ops.tokens_to_quads(q, *hw0), ops.tokens_to_quads(k, *hw1), ops.tokens_to_quads(v, *hw1),
                                   ops.topk_idx_to_tab(prev), hw0, hw1, H, topk, **kw)


def test_quad_major_layout_passes(ops):
    """NCHW -> quad-major, tokens -> quad-major and the int64 -> int32 table transposition are pure data movement"""
    r = np.random.default_rng(5)
    shapes = [(2, 64, 12, 20), (2, 256, 6, 70), (2, 32, 2, 2), (2, 128, 26, 26)]
    xs = [r.standard_normal(sh).astype(np.float32) for sh in shapes]

    def ref(x):
        B, C, h, w = x.shape
        y = x.reshape(B, C // 32, 32, h // 2, 2, w // 2, 2).transpose(0, 1, 3, 5, 4, 6, 2)   # b hd qy qx r c d
        return np.ascontiguousarray(y).reshape(B, C // 32, (h // 2) * (w // 2), 4, 32)
    outs = ops.nchw_to_quads_multi([T(x) for x in xs])
    for x, o in zip(xs, outs):
        assert np.array_equal(N(o), ref(x))
        B, C, h, w = x.shape
        assert np.array_equal(N(ops.tokens_to_quads(T(_tok(x)), h, w)), ref(x))
    # one launch, mixed: token-major for some tensors (the coarsest level of a QTAttB call, odd grids allowed), quad-major for the others
    odd = r.standard_normal((2, 96, 13, 7)).astype(np.float32)
    mixed = ops.nchw_to_quads_multi([T(xs[0]), T(odd), T(xs[1]), T(xs[3])], tokens=[True, True, False, True])
    assert np.array_equal(N(mixed[0]), _tok(xs[0])) and np.array_equal(N(mixed[1]), _tok(odd))
    assert np.array_equal(N(mixed[2]), ref(xs[1])) and np.array_equal(N(mixed[3]), _tok(xs[3]))
    idx = r.integers(0, 1000, (3, 17, 5, 4)).astype(np.int64)
    assert np.array_equal(N(ops.topk_idx_to_tab(T(idx))), idx.transpose(0, 3, 1, 2).astype(np.int32))


@pytest.mark.parametrize("kernel", ["qm", "dma", "quad"])
@pytest.mark.parametrize("name", list(CASES["qtattb"]))
def test_qtattb_levels(ops, monkeypatch, name, kernel):
    """coarse + fine level kernels chained exactly like QTAttB.forward; indices bit-exact vs oracle AND vs the reference."""
    monkeypatch.setenv("CASMTR_FINE_KERNEL", kernel)   # register-value kernel (lists <= 64) | LDS-DMA kernel | round-1 workgroup-per-quad kernel
    inp = make_inputs("qtattb", name)
    cfg = CASES["qtattb"][name]
    H, topks = cfg["nhead"], cfg["topks"]
    final_o, lv_o = oracle.qtattb_forward(inp["queries"], inp["keys"], inp["values"], inp["weight"], H, topks)
    g = load_golden("qtattb", name)
    w = np.asarray(inp["weight"], np.float32)
    e = np.exp(w - w.max()); wsm = (e / e.sum()).astype(np.float32)
    acc = prev = None
    for i in range(3):
        q, k, v = (_tok(inp[n][2 - i]) for n in ("queries", "keys", "values"))
        h0, w0 = inp["queries"][2 - i].shape[2:]
        h1, w1 = inp["keys"][2 - i].shape[2:]
        if i == 0:
            out = ops.qta_coarse_level(T(q), T(k), T(v), H, topks[0], w_level=float(wsm[0]))
        else:
            out = _fine_level(ops, kernel, T(q), T(k), T(v), prev, (h0, w0), (h1, w1), H, topks[i], w_level=float(wsm[i]), acc_in=acc)
        acc, prev = out["acc"], out["topk_idx"]
        assert np.array_equal(N(out["topk_idx"]), lv_o[i]["topk_idx"]), f"level {i} top-k indices differ from the oracle"
        assert np.array_equal(N(out["topk_idx"]), g[f"L{i}_topk_idx"].astype(np.int64)), f"level {i} top-k differ from the reference"
        assert_close(N(out["topk_score"]), lv_o[i]["topk_score"], SOFTMAX_TOL, f"level {i} topk_score")
        assert_close(N(out["message"]), lv_o[i]["message"], SOFTMAX_TOL, f"level {i} message")
    assert_close(N(acc), final_o, SOFTMAX_TOL, "final message vs oracle")
    assert_close(N(acc), g["final"], SOFTMAX_TOL, "final message vs reference python")


@pytest.mark.parametrize("kernel", ["dma", "quad"])
@pytest.mark.parametrize("name", list(CASES["cascade_attn"]))
def test_cascade_attn(ops, monkeypatch, name, kernel):
    monkeypatch.setenv("CASMTR_CASCADE_KERNEL", kernel)   # default persistent LDS-DMA + MFMA kernel | round-1 workgroup-per-quad kernel
    inp = make_inputs("cascade_attn", name)
    cfg = CASES["cascade_attn"][name]
    g = load_golden("cascade_attn", name)
    hc, wc = cfg["coarse_hw"]
    tp = ops.window_warp_idx(T(inp["coarse_idx"]), hc, wc, cfg["ws"])
    assert np.array_equal(N(tp), g["topk_pos"].astype(np.int64))
    h, w = 2 * hc, 2 * wc
    rel = T(inp["rel_pos"]) if cfg.get("rel_pos") else None
    q, k, v = (ops.nchw_to_tokens(T(inp[n])) for n in ("q", "k", "v"))
    msg, up = ops.cascade_attn(q, k, v, tp, (h, w), (h, w), cfg["nhead"], rel_pos=rel)
    assert np.array_equal(N(up), g["upsampled_idx"].astype(np.int64))
    assert_close(N(msg), g["message"], SOFTMAX_TOL, "message vs reference python")
    mo, _ = oracle.cascade_attn(N(q), N(k), N(v), N(tp), (h, w), (h, w), cfg["nhead"], rel_pos=inp.get("rel_pos"))
    assert_close(N(msg), mo, 1e-5, "message vs oracle")


@pytest.mark.parametrize("name", list(CASES["cascade_attn"]))
def test_cascade_attn_quad_major_vs_reference(ops, name):
    """the round-3 cascade kernel (quad-major operands, quad pairs sharing a window box) on the reference fixtures"""
    inp = make_inputs("cascade_attn", name)
    cfg = CASES["cascade_attn"][name]
    g = load_golden("cascade_attn", name)
    hc, wc = cfg["coarse_hw"]
    tp = ops.window_warp_idx(T(inp["coarse_idx"]), hc, wc, cfg["ws"])
    h, w = 2 * hc, 2 * wc
    rel = T(inp["rel_pos"]) if cfg.get("rel_pos") else None
    assert ops.cascade_quad_supported(cfg["nhead"], 32, (h, w), (h, w), tp.shape[2], 1)
    qm = ops.nchw_to_quads_multi([T(inp[n]) for n in ("q", "k", "v")])
    msg = ops.cascade_attn_quad(qm[0], qm[1], qm[2], tp, (h, w), (h, w), cfg["nhead"], rel_pos=rel)
    assert_close(N(msg), g["message"], SOFTMAX_TOL, "message vs reference python")
    q, k, v = (_tok(inp[n]) for n in ("q", "k", "v"))
    mo, _ = oracle.cascade_attn(q, k, v, N(tp), (h, w), (h, w), cfg["nhead"], rel_pos=inp.get("rel_pos"))
    assert_close(N(msg), mo, 2e-5, "message vs oracle")


@pytest.mark.parametrize("kind", ["smooth", "random", "identical", "two_apart", "irregular", "edges"])
@pytest.mark.parametrize("H,hc,wc,with_rel", [(4, 12, 14, False), (2, 10, 9, True), (8, 6, 7, False), (1, 8, 10, True)])
def test_cascade_attn_quad_major_window_modes(ops, kind, H, hc, wc, with_rel):
    """every way two horizontally adjacent quads' windows can relate: one column apart (shared 5 x 6 box), identical (5 x 5 box), two
    columns apart / different rows / random (two single-quad sub-items), irregular position lists (not a 5 x 5 block: cells listed
    one by one), windows pushed against the grid's edges, odd numbers of quads per row (last quad alone), every head count / XCD
    split, several pairs, with and without rel_pos -- message against the oracle"""
    r = np.random.default_rng(sum(map(ord, kind)) + 100 * H + hc)
    B = 2
    h, w, C = 2 * hc, 2 * wc, H * 32
    q, k, v = (r.standard_normal((B, h * w, C)).astype(np.float32) for _ in range(3))
    ys, xs = np.meshgrid(np.arange(hc), np.arange(wc), indexing="ij")
    if kind == "smooth":
        cy, cx = np.clip(ys + 1, 0, hc - 1), np.clip(xs + 2, 0, wc - 1)
    elif kind == "random":
        cy, cx = r.integers(0, hc, (hc, wc)), r.integers(0, wc, (hc, wc))
    elif kind == "identical":
        cy, cx = np.clip(ys, 0, hc - 1), np.clip((xs // 2) * 2 + 1, 0, wc - 1)
    elif kind == "two_apart":
        cy, cx = ys, np.clip(xs * 2 - wc // 2, 0, wc - 1)
    elif kind == "edges":
        cy, cx = np.where(ys < hc // 2, 0, hc - 1), np.where(xs < wc // 2, 0, wc - 1)
    else:
        cy, cx = np.clip(ys + 1, 0, hc - 1), np.clip(xs + 2, 0, wc - 1)
    cidx = np.broadcast_to((cy * wc + cx).reshape(1, -1), (B, hc * wc)).astype(np.int64).copy()
    tp = oracle.window_warp_idx(cidx, hc, wc, 5)
    if kind == "irregular":   # arbitrary in-range cells for some quads: still a valid topk_pos for the op
        sel = r.random((B, hc * wc)) < 0.3
        tp[sel] = np.stack([r.integers(0, hc, (int(sel.sum()), 25)), r.integers(0, wc, (int(sel.sum()), 25))], -1)
    rel = r.standard_normal((B, H, h * w, 100)).astype(np.float32) if with_rel else None
    mo, _ = oracle.cascade_attn(q, k, v, tp, (h, w), (h, w), H, rel_pos=rel)
    qm = [ops.tokens_to_quads(T(x), h, w) for x in (q, k, v)]
    msg = ops.cascade_attn_quad(qm[0], qm[1], qm[2], T(tp), (h, w), (h, w), H, rel_pos=None if rel is None else T(rel))
    assert_close(N(msg), mo, 2e-5, f"{kind}: message vs oracle")
    if H > 1:   # (the token-major kernels cover 2, 4 and 8 heads)
        msg_old, _ = ops.cascade_attn(T(q), T(k), T(v), T(tp), (h, w), (h, w), H, rel_pos=None if rel is None else T(rel), want_idx=False)
        assert_close(N(msg), N(msg_old), 2e-5, f"{kind}: message vs the token-major kernel")


def _valid_hw(m0, m1):
    return np.stack([m0.sum(1).max(-1), m0.sum(2).max(-1), m1.sum(1).max(-1), m1.sum(2).max(-1)], 1).astype(np.int32)


@pytest.mark.parametrize("gemm", ["split", "exact"])
@pytest.mark.parametrize("recip", [False, True])
@pytest.mark.parametrize("name", list(CASES["coarse_matching"]))
def test_dual_softmax(ops, name, recip, gemm):
    inp = make_inputs("coarse_matching", name)
    cfg = CASES["coarse_matching"][name]
    B = cfg["B"]
    kw = dict(temperature=cfg.get("T", 0.1), thr=cfg.get("thr", 0.2), border_rm=cfg.get("border_rm", 0))
    m0 = m1 = valid = None
    if cfg.get("masks"):
        m0, m1 = inp["mask0"].reshape(B, -1), inp["mask1"].reshape(B, -1)
        valid = _valid_hw(inp["mask0"], inp["mask1"])
    o = oracle.dual_softmax(inp["feat0"], inp["feat1"], cfg["hw0"], cfg["hw1"], mask0=m0, mask1=m1, valid_hw=valid,
                            recip=recip, want_conf=True, **kw)
    d = ops.dual_softmax(T(inp["feat0"]), T(inp["feat1"]), cfg["hw0"], cfg["hw1"], mask0=None if m0 is None else T(m0),
                         mask1=None if m1 is None else T(m1), valid_hw=None if valid is None else T(valid), recip=recip,
                         want_conf=True, gemm=gemm, **kw)
    assert np.array_equal(N(d["next_idx_c01"]), o["next_idx_c01"]), "row argmax must be bit-exact"
    assert np.array_equal(N(d["next_idx_c10"]), o["next_idx_c10"]), "column argmax must be bit-exact"
    assert_close(N(d["next_conf_c01"]), o["next_conf_c01"], SOFTMAX_TOL, "next_conf_c01")
    assert_close(N(d["next_conf_c10"]), o["next_conf_c10"], SOFTMAX_TOL, "next_conf_c10")
    assert_close(N(d["conf_matrix"]), o["conf_matrix"], SOFTMAX_TOL, "conf_matrix")
    n = int(d["n"].item())
    got = match_set(N(d["b_ids"][:n]), N(d["i_ids"][:n]), N(d["j_ids"][:n]))
    want = match_set(o["b_ids"], o["i_ids"], o["j_ids"])
    for t in got ^ want:  # only threshold-borderline entries may differ (expf differs by ulps between CPU and GPU)
        assert abs(o["conf_matrix"][t] - kw["thr"]) < 1e-5, f"match list differs at {t}"
    assert len(got ^ want) <= 1
    if got == want:
        assert np.array_equal(N(d["i_ids"][:n]), o["i_ids"]) and np.array_equal(N(d["j_ids"][:n]), o["j_ids"]), "order"
        assert_close(N(d["mconf"][:n]), o["mconf"], SOFTMAX_TOL, "mconf")
    if gemm == "split":   # exact by construction: the SAME list as the all-fp32 path (the <= 1 above is device exp against libm's)
        ex = ops.dual_softmax(T(inp["feat0"]), T(inp["feat1"]), cfg["hw0"], cfg["hw1"], mask0=None if m0 is None else T(m0),
                              mask1=None if m1 is None else T(m1), valid_hw=None if valid is None else T(valid), recip=recip,
                              want_conf=True, gemm="exact", **kw)
        assert int(ex["n"].item()) == n and torch.equal(ex["i_ids"][:n], d["i_ids"][:n]) and torch.equal(ex["j_ids"][:n], d["j_ids"][:n]) \
            and torch.equal(ex["b_ids"][:n], d["b_ids"][:n]), "split and exact GEMM paths must return the same match list"
    if not recip:  # the fixtures come from the reference on CPU (true division)
        g = load_golden("coarse_matching", name)
        audit_index_mismatches(N(d["next_idx_c01"]), g["next_idx_c01"], dot_score_fn(inp["feat0"], inp["feat1"], m0, m1), "next_idx_c01 vs reference")
        audit_index_mismatches(N(d["next_idx_c10"]), g["next_idx_c10"], dot_score_fn(inp["feat1"], inp["feat0"], m1, m0), "next_idx_c10 vs reference")
        assert_close(N(d["next_conf_c01"]), g["next_conf_c01"], SOFTMAX_TOL, "next_conf_c01 vs reference python")
        gs = match_set(g["b_ids"], g["i_ids"], g["j_ids"])
        assert len(got ^ gs) <= 1


@pytest.mark.parametrize("recip", [False, True])
@pytest.mark.parametrize("name", list(CASES["cascade_matching"]))
def test_window_match_and_select(ops, name, recip):
    inp = make_inputs("cascade_matching", name)
    cfg = CASES["cascade_matching"][name]
    g = load_golden("cascade_matching", name)
    hc, wc = cfg["coarse_hw"]
    h, w = 2 * hc, 2 * wc
    B = cfg["B"]
    idx01, idx10 = g["idx_c01"].astype(np.int64), g["idx_c10"].astype(np.int64)
    mq = mk = valid = None
    if cfg.get("masks"):
        mq, mk = inp["mask0"].reshape(B, -1), inp["mask1"].reshape(B, -1)
        valid = _valid_hw(inp["mask0"], inp["mask1"])
    tm = lambda m: None if m is None else T(m)
    for quad in (True, False):
        d01 = ops.window_match(T(inp["feat0"]), T(inp["feat1"]), T(idx01), 1.0, tm(mq), tm(mk), recip=recip, hw=(h, w) if quad else None)
        d10 = ops.window_match(T(inp["feat1"]), T(inp["feat0"]), T(idx10), 1.0, tm(mk), tm(mq), recip=recip, hw=(h, w) if quad else None)
        o01 = oracle.window_match(inp["feat0"], inp["feat1"], idx01, 1.0, mq, mk, recip=recip)
        o10 = oracle.window_match(inp["feat1"], inp["feat0"], idx10, 1.0, mk, mq, recip=recip)
        assert np.array_equal(N(d01["next_idx"]), o01["next_idx"]), "next_idx_c01 must be bit-exact"
        assert np.array_equal(N(d10["next_idx"]), o10["next_idx"]), "next_idx_c10 must be bit-exact"
        assert_close(N(d01["conf_matrix"]), o01["conf_matrix"], SOFTMAX_TOL, "conf_matrix01")
        assert_close(N(d01["next_conf"]), o01["next_conf"], SOFTMAX_TOL, "next_conf_c01")
        assert_close(N(d10["next_conf"]), o10["next_conf"], SOFTMAX_TOL, "next_conf_c10")
    # generic (non-identical) index rows exercise the per-token restaging path
    r = np.random.default_rng(5)
    ridx = r.integers(0, h * w, idx01.shape, dtype=np.int64)
    dg = ops.window_match(T(inp["feat0"]), T(inp["feat1"]), T(ridx), 1.0, recip=recip, hw=(h, w))
    og = oracle.window_match(inp["feat0"], inp["feat1"], ridx, 1.0, recip=recip)
    assert np.array_equal(N(dg["next_idx"]), og["next_idx"])
    assert_close(N(dg["conf_matrix"]), og["conf_matrix"], SOFTMAX_TOL, "generic conf")
    # selection: fed with the reference's own stage outputs it is pure comparison logic -> exact vs the fixture
    post, extra = cfg.get("post"), None
    if post:   # 'local_window_nms' / 'd2d': the survivors arrive as an extra keep mask (here: the oracle's restatement of the method)
        extra = T(post_extra_mask(post, g["next_conf_c01"], inp["feat0"], (h, w)).astype(np.uint8))
    sel = ops.nms_select(T(g["next_conf_c01"]), T(g["next_idx_c01"].astype(np.int64)), T(g["next_idx_c10"].astype(np.int64)),
                         (h, w), (h, w), nms_window=5 if (cfg.get("nms", True) and not post) else 0, test_thr=cfg.get("test_thr", 0.2),
                         pre=[(T(inp["pre_conf"]), (hc, wc), cfg.get("pre_thr", 0.2))], border_rm=cfg.get("border_rm", 2),
                         valid_hw=None if valid is None else T(valid), double_check=cfg.get("double_check", True), extra_keep=extra)
    n = int(sel["n"].item())
    assert n == len(g["b_ids"])
    assert np.array_equal(N(sel["b_ids"][:n]), g["b_ids"].astype(np.int64))
    assert np.array_equal(N(sel["i_ids"][:n]), g["i_ids"].astype(np.int64))
    assert np.array_equal(N(sel["j_ids"][:n]), g["j_ids"].astype(np.int64))
    assert np.array_equal(N(sel["mconf"][:n]), g["mconf"])


def test_nms_keep_one_fallback(ops):
    B, h, w = 3, 8, 8
    conf = torch.zeros((B, h * w), device=DEV)
    idx = torch.arange(h * w, device=DEV).repeat(B, 1)
    sel = ops.nms_select(conf, idx, idx, (h, w), (h, w), nms_window=5, test_thr=0.2)
    n = int(sel["n"].item())
    assert n == B and N(sel["b_ids"][:n]).tolist() == [0, 1, 2] and N(sel["i_ids"][:n]).tolist() == [0, 0, 0]
    o = oracle.nms_select(N(conf), N(idx), N(idx), (h, w), (h, w), nms_window=5, test_thr=0.2)
    assert o["b_ids"].tolist() == [0, 1, 2]


def test_contract_errors(ops):
    q = torch.zeros((1, 4, 4, 2, 32), device=DEV)
    key = torch.zeros((1, 16, 2, 32), device=DEV)
    idx = torch.zeros((1, 4, 8, 2), device=DEV, dtype=torch.int64)
    with pytest.raises(RuntimeError):
        ops.qta_score_fwd(q.cpu(), key, idx)            # not device resident (score_computation.cpp:6)
    with pytest.raises(RuntimeError):
        ops.qta_score_fwd(q.transpose(1, 2), key, idx)   # not contiguous (score_computation.cpp:7)
    with pytest.raises(RuntimeError):
        ops.qta_score_fwd(q, key, idx.int())            # index dtype


@pytest.mark.parametrize("H,Kp,topk", [(4, 40, 24), (2, 64, 16), (8, 33, 8)])
def test_fine_level_wide_candidate_lists(ops, H, Kp, topk):
    """K = 4*Kp in (128, 256]: the KMAX = 256 instantiations (4 systolic passes per head), every head count -- bit-exact top-k"""
    r = np.random.default_rng(100 * H + Kp)
    B, (h0, w0), (h1, w1) = 2, (12, 16), (24, 16)
    C = H * 32
    q = r.standard_normal((B, h0 * w0, C)).astype(np.float32)
    k = r.standard_normal((B, h1 * w1, C)).astype(np.float32)
    v = r.standard_normal((B, h1 * w1, C)).astype(np.float32)
    Lq, Sp = (h0 // 2) * (w0 // 2), (h1 // 2) * (w1 // 2)
    prev = np.stack([np.stack([r.permutation(Sp)[:Kp] for _ in range(H)], -1) for _ in range(B * Lq)]).reshape(B, Lq, Kp, H)
    acc_in = r.standard_normal((B, Lq, H, 32)).astype(np.float32)
    o = oracle.qta_fine_level(q.reshape(B, -1, H, 32), k.reshape(B, -1, H, 32), v.reshape(B, -1, H, 32), prev, (h0, w0), (h1, w1),
                              topk, 0.37, acc_in)
    out = ops.qta_fine_level(T(q), T(k), T(v), T(prev.astype(np.int64)), (h0, w0), (h1, w1), H, topk, w_level=0.37, acc_in=T(acc_in))
    assert np.array_equal(N(out["topk_idx"]), o["topk_idx"])
    assert_close(N(out["topk_score"]), o["topk_score"], SOFTMAX_TOL, "topk_score")
    assert_close(N(out["acc"]), o["acc"], SOFTMAX_TOL, "merged message")


@pytest.mark.parametrize("H,Kp,topk,with_acc", [(8, 16, 8, True), (8, 32, 16, True), (8, 16, 0, True), (4, 32, 16, False), (2, 9, 5, True),
                                                 (1, 16, 4, True), (4, 5, 20, True), (8, 1, 4, True), (2, 13, 0, False)])
@pytest.mark.parametrize("kernel", ["qm", "dma"])
def test_fine_level_dma_kernel_shapes(ops, monkeypatch, kernel, H, Kp, topk, with_acc):
    """fine_level_dma_kernel (K <= 128) against the quad-major kernel: every head count /
    XCD split, ragged candidate counts, top-k == K, no top-k (finest level), no incoming accumulator, query grid != key grid,
    several pairs -- top-k bit-exact, messages within tolerance"""
    monkeypatch.setenv("CASMTR_FINE_KERNEL", kernel)
    if kernel == "qm" and topk > 16:
        pytest.skip("the quad-major kernel keeps the 16 best (every shipped config); longer lists run the token-major kernels")
    r = np.random.default_rng(1000 * H + 10 * Kp + topk)
    B, (h0, w0), (h1, w1) = 3, (12, 20), (16, 12)
    C = H * 32
    q = r.standard_normal((B, h0 * w0, C)).astype(np.float32)
    k = r.standard_normal((B, h1 * w1, C)).astype(np.float32)
    v = r.standard_normal((B, h1 * w1, C)).astype(np.float32)
    Lq, Sp = (h0 // 2) * (w0 // 2), (h1 // 2) * (w1 // 2)
    prev = np.stack([np.stack([r.permutation(Sp)[:Kp] for _ in range(H)], -1) for _ in range(B * Lq)]).reshape(B, Lq, Kp, H)
    acc_in = r.standard_normal((B, Lq, H, 32)).astype(np.float32) if with_acc else None
    o = oracle.qta_fine_level(q.reshape(B, -1, H, 32), k.reshape(B, -1, H, 32), v.reshape(B, -1, H, 32), prev, (h0, w0), (h1, w1),
                              topk, 0.37, acc_in)
    out = _fine_level(ops, kernel, T(q), T(k), T(v), T(prev.astype(np.int64)), (h0, w0), (h1, w1), H, topk, w_level=0.37,
                      acc_in=None if acc_in is None else T(acc_in))
    if topk:
        assert np.array_equal(N(out["topk_idx"]), o["topk_idx"])
        assert_close(N(out["topk_score"]), o["topk_score"], SOFTMAX_TOL, "topk_score")
        if kernel == "qm":   # the compact table the next level reads holds the same indices
            assert np.array_equal(N(out["topk_tab"]), o["topk_idx"].transpose(0, 3, 1, 2).astype(np.int32))
    assert_close(N(out["message"]), o["message"], SOFTMAX_TOL, "message")
    assert_close(N(out["acc"]), o["acc"], SOFTMAX_TOL, "merged message")


@pytest.mark.parametrize("kind", ["duplicate_rows", "near_ties", "all_equal", "one_bucket"])
@pytest.mark.parametrize("Kp,topk", [(32, 16), (16, 8), (7, 16)])
def test_fine_quad_selection_ties(ops, kind, Kp, topk):
    """The quad-major kernel selects on packed 32-bit keys (25 high bits of the ordered logit | position) and must fall back to the
    exact iterated argmax whenever two of the best logits share those bits: exact ties (identical key rows: first position wins, as
    torch.topk on the oracle's order), logits a few ulp apart, everything equal, all logits inside one 128-ulp bucket."""
    r = np.random.default_rng(sum(map(ord, kind)) * 1000 + 10 * Kp + topk)
    B, H, (h0, w0), (h1, w1) = 2, 4, (8, 12), (16, 16)
    C = H * 32
    q = r.standard_normal((B, h0 * w0, C)).astype(np.float32)
    k = r.standard_normal((B, h1 * w1, C)).astype(np.float32)
    v = r.standard_normal((B, h1 * w1, C)).astype(np.float32)
    if kind == "duplicate_rows":      # many identical key rows -> exact ties among the candidates
        k = k.reshape(B, -1, H, 32); k[:, 1::2] = k[:, 0::2]; k[:, 5::8] = k[:, 0:1]; k = k.reshape(B, -1, C)
    elif kind == "near_ties":         # rows that differ in the last bits only
        k = k.reshape(B, -1, H, 32); base = k[:, :1].copy()
        k[:] = base * (1.0 + 1e-7 * r.integers(-3, 4, (B, h1 * w1, H, 1)).astype(np.float32)); k = k.reshape(B, -1, C)
    elif kind == "all_equal":
        k[:] = k[:, :1]
    elif kind == "one_bucket":        # tiny keys: every logit lies within a few ulp of a constant set by one large channel
        k = (1e-6 * k).reshape(B, -1, H, 32); k[..., 0] = 1.0; k = k.reshape(B, -1, C)
        q = q.reshape(B, -1, H, 32); q[..., 0] = 3.0; q = q.reshape(B, -1, C)
    Lq, Sp = (h0 // 2) * (w0 // 2), (h1 // 2) * (w1 // 2)
    prev = np.stack([np.stack([r.permutation(Sp)[:Kp] for _ in range(H)], -1) for _ in range(B * Lq)]).reshape(B, Lq, Kp, H)
    tk = min(topk, 4 * Kp)
    o = oracle.qta_fine_level(q.reshape(B, -1, H, 32), k.reshape(B, -1, H, 32), v.reshape(B, -1, H, 32), prev, (h0, w0), (h1, w1),
                              tk, 0.5, None)
    out = _fine_level(ops, "qm", T(q), T(k), T(v), T(prev.astype(np.int64)), (h0, w0), (h1, w1), H, tk, w_level=0.5)
    assert np.array_equal(N(out["topk_idx"]), o["topk_idx"]), f"{kind}: top-k indices"
    assert_close(N(out["topk_score"]), o["topk_score"], SOFTMAX_TOL, "topk_score")
    assert_close(N(out["acc"]), o["acc"], SOFTMAX_TOL, "merged message")


@pytest.mark.parametrize("H,ws", [(2, 7), (4, 7), (8, 3)])
def test_cascade_attn_other_windows(ops, H, ws):
    """window 7 -> K = 196 (KMAX = 256), window 3 -> K = 36 (KMAX = 64); heads 2 / 4 / 8; rel_pos on"""
    r = np.random.default_rng(7 * H + ws)
    B, hc, wc = 1, 10, 8
    h, w, C = 2 * hc, 2 * wc, H * 32
    q, k, v = (r.standard_normal((B, h * w, C)).astype(np.float32) for _ in range(3))
    cidx = r.integers(0, hc * wc, (B, hc * wc))
    tp = oracle.window_warp_idx(cidx, hc, wc, ws)
    rel = r.standard_normal((B, H, h * w, 4 * ws * ws)).astype(np.float32)
    mo, uo = oracle.cascade_attn(q, k, v, tp, (h, w), (h, w), H, rel_pos=rel)
    tpg = ops.window_warp_idx(T(cidx.astype(np.int64)), hc, wc, ws)
    assert np.array_equal(N(tpg), tp)
    msg, up = ops.cascade_attn(T(q), T(k), T(v), tpg, (h, w), (h, w), H, rel_pos=T(rel))
    assert np.array_equal(N(up), uo)
    assert_close(N(msg), mo, 2e-5, "message vs oracle")


def test_empty_batch_and_limits(ops):
    """B = 0 returns empty tensors (no launch); shapes outside the kernels' envelope raise instead of falling back"""
    z = lambda *s: torch.zeros(s, device=DEV)
    out = ops.qta_coarse_level(z(0, 16, 64), z(0, 16, 64), z(0, 16, 64), 2, 4, w_level=1.0)
    assert out["acc"].shape[0] == 0 and out["topk_idx"].shape == (0, 16, 4, 2)
    f = ops.dual_softmax(z(0, 16, 32), z(0, 16, 32), (4, 4), (4, 4), 0.1, 0.2)
    assert int(f["n"].item()) == 0 if "n" in f else True
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):   # odd grid side below a finer level (reference: view() would fail)
        ops.qta_fine_level(z(1, 15, 64), z(1, 16, 64), z(1, 16, 64), torch.zeros((1, 3, 2, 2), device=DEV, dtype=torch.int64),
                           (3, 5), (4, 4), 2, 2)
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):   # more than 256 candidates per quad
        ops.qta_fine_level(z(1, 16, 64), z(1, 1600, 64), z(1, 1600, 64), torch.zeros((1, 4, 80, 2), device=DEV, dtype=torch.int64),
                           (4, 4), (40, 40), 2, 2)
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):   # window list longer than 128
        ops.window_match(z(1, 16, 64), z(1, 400, 64), torch.zeros((1, 16, 144), device=DEV, dtype=torch.int64))


@pytest.mark.parametrize("kernel", ["tile", "three"])
@pytest.mark.parametrize("kind", ["random", "all_equal", "one_lane_heavy", "many_ties", "few_valid", "wide_range", "indoor", "ragged", "tiny", "widest", "one_head"])
def test_coarse_topk_paths(ops, monkeypatch, kind, kernel):
    """coarse-level top-k: the sorted fast path (<= 64 survivors of the lane-maxima threshold; in the tile kernel the exact 32-bit
    packing) and the fallbacks (ties / concentrated rows: iterated argmax; survivors spread over many binades: pair sort) must all
    return the oracle's list, ordered (logit desc, position asc)"""
    monkeypatch.setenv("CASMTR_COARSE_KERNEL", kernel)   # round-4 register-tile kernel (default) | logits / row / A.V kernels | LDS-tile kernel
    r = np.random.default_rng({"random": 1, "all_equal": 2, "one_lane_heavy": 3, "many_ties": 4, "few_valid": 5, "wide_range": 6,
                               "indoor": 7, "ragged": 8, "tiny": 9, "widest": 10, "one_head": 11}[kind])
    B, H, L, S, topk = 1, 2, 40, 676, 32
    if kind == "few_valid":
        S, topk = 48, 8          # fewer keys than lanes
    elif kind == "indoor":
        B, H, L, S = 2, 8, 300, 300   # 20 x 15 grid (configs[4]): 5 key blocks, the last one partial, 19 row tiles
    elif kind == "ragged":
        L, S, topk = 37, 131, 16      # nothing a multiple of anything
    elif kind == "tiny":
        L, S, topk = 16, 16, 8        # 4 x 4 coarsest grid of the small fixtures
    elif kind == "widest":
        H, L, S, topk = 4, 33, 1000, 60   # 16 key blocks (the largest register tile), the largest list the tile kernel takes
    elif kind == "one_head":
        H, L, S, topk = 1, 50, 200, 16
    q = r.standard_normal((B, L, H, 32)).astype(np.float32)
    k = r.standard_normal((B, S, H, 32)).astype(np.float32)
    v = r.standard_normal((B, S, H, 32)).astype(np.float32)
    if kind == "all_equal":
        k[:] = k[:, :1]                      # every key identical -> every logit of a row identical
    elif kind == "one_lane_heavy":
        k[:, 5::64] = 3.0 * q[:, :1].mean(axis=1, keepdims=True) + k[:, 5::64]   # positions 5, 69, 133, ... share lane 5
        q[:] = q[:, :1]                      # same query everywhere so that the boost lines up
    elif kind == "many_ties":
        k[:, ::2] = k[:, 1::2]               # every logit appears twice
    elif kind == "wide_range":
        # the 40 best logits of a row span 20+ binades (ordered keys further apart than 2^26): keys = query direction x 2^-j
        k[:, :40] = q[:, :1].mean(axis=1, keepdims=True) * (2.0 ** -np.arange(40, dtype=np.float32))[None, :, None, None] * 8.0
        k[:, 40:] *= 1e-9
        q[:] = q[:, :1]
    C = H * 32
    o = oracle.qta_coarse_level(q, k, v, topk)
    out = ops.qta_coarse_level(T(q.reshape(B, L, C)), T(k.reshape(B, S, C)), T(v.reshape(B, S, C)), H, topk, w_level=1.0, want_tab=True)
    assert np.array_equal(N(out["topk_idx"]), o[2]), kind
    assert np.array_equal(N(out["topk_tab"]), o[2].transpose(0, 3, 1, 2)), "int32 table [B,H,L,topk]"
    assert_close(N(out["topk_score"]), o[1], SOFTMAX_TOL, "topk_score")
    assert_close(N(out["message"]), o[0], SOFTMAX_TOL, "message")
    assert_close(N(out["acc"]), o[0], SOFTMAX_TOL, "message * weight")
    if kernel == "tile":   # the lists on request only: table-only call, same table
        out2 = ops.qta_coarse_level(T(q.reshape(B, L, C)), T(k.reshape(B, S, C)), T(v.reshape(B, S, C)), H, topk, w_level=0.5,
                                    want_message=False, want_tab=True, want_topk=False)
        assert out2["topk_idx"] is None and out2["message"] is None
        assert np.array_equal(N(out2["topk_tab"]), o[2].transpose(0, 3, 1, 2))
        assert_close(N(out2["acc"]), o[0] * np.float32(0.5), SOFTMAX_TOL, "message * 0.5")


@pytest.mark.parametrize("kernel", ["quad", "dma", "pair"])
@pytest.mark.parametrize("recip", [False, True])
@pytest.mark.parametrize("C,ws,masks,dil", [(128, 5, False, 1), (128, 5, True, 1), (64, 5, False, 1), (128, 3, False, 1),
                                            (256, 5, False, 1), (32, 5, True, 2)])
def test_window_match_implicit_windows(ops, monkeypatch, C, ws, masks, dil, recip, kernel):
    """casmtr_window_match_pos_fwd (topk_pos in, candidates expanded in-kernel, LDS-DMA key staging) == the explicit-index
    kernel == the oracle on the expanded tensor; casmtr_window_expand_idx == CascadeQTAttB's upsampled_idx."""
    monkeypatch.setenv("CASMTR_WINDOW_KERNEL", kernel)   # round-1 wave-per-quad kernel | persistent LDS-DMA + MFMA kernel, one quad per item |
                                                         # default: quad pairs on shared window boxes where the shape allows it
    B, hc, wc = 2, 12, 16
    h, w = 2 * hc, 2 * wc
    r = np.random.default_rng(100 + C + ws)
    cidx = r.integers(0, hc * wc, (B, hc * wc), dtype=np.int64)
    tp = ops.window_warp_idx(T(cidx), hc, wc, ws)
    wi = ops.WindowIndex(tp, (h, w), (h, w), dil)
    z = np.zeros((B, h * w, 32), np.float32)
    _, up_o = oracle.cascade_attn(z, z, z, N(tp), (h, w), (h, w), 1, dilated=dil)
    assert np.array_equal(N(wi.materialize()), up_o), "expanded window indices"
    fq = 2.0 * r.standard_normal((B, h * w, C), dtype=np.float32)
    fk = fq + 0.7 * r.standard_normal((B, h * w, C), dtype=np.float32)
    mq = mk = None
    if masks:
        mq = (r.random((B, h * w)) > 0.2).astype(np.uint8)
        mk = (r.random((B, h * w)) > 0.2).astype(np.uint8)
        mq[0, :40] = 0   # fully masked query rows: uniform softmax, argmax = candidate 0
    kw = dict(mask_q=None if mq is None else T(mq), mask_k=None if mk is None else T(mk), recip=recip)
    d = ops.window_match(T(fq), T(fk), wi, 1.0, **kw)
    e = ops.window_match(T(fq), T(fk), wi.materialize(), 1.0, hw=(h, w), **kw)
    o = oracle.window_match(fq, fk, up_o, 1.0, mq, mk, recip=recip)
    assert np.array_equal(N(d["next_idx"]), o["next_idx"]), "argmax must be bit-exact vs the oracle"
    assert torch.equal(d["next_idx"], e["next_idx"]) and torch.equal(d["conf_matrix"], e["conf_matrix"]) and torch.equal(d["next_conf"], e["next_conf"])
    assert_close(N(d["conf_matrix"]), o["conf_matrix"], SOFTMAX_TOL, "conf_matrix")
    assert_close(N(d["next_conf"]), o["next_conf"], SOFTMAX_TOL, "next_conf")
    n = ops.window_match(T(fq), T(fk), wi, 1.0, want_conf=False, **kw)
    assert n["conf_matrix"] is None and torch.equal(n["next_idx"], d["next_idx"])


@pytest.mark.parametrize("C,hc,wc,masks,conf", [(128, 14, 16, False, True), (128, 13, 15, True, True), (64, 16, 21, False, True), (64, 12, 12, True, False)])
def test_window_match_pair_kernel(ops, monkeypatch, C, hc, wc, masks, conf):
    """window_match_pair_kernel (two quads per item on a shared 5 x 5 / 5 x 6 box) == window_match_pos_kernel (one quad per item) bit for
    bit, == the oracle's argmax, on windows that follow a smooth coarse match field (most pairs share a box), with jumps, an irregular
    position list and an odd number of quads per row (the last quad of a row runs alone)."""
    B = 2
    h, w = 2 * hc, 2 * wc
    r = np.random.default_rng(7 + C + hc)
    yy, xx = np.meshgrid(np.arange(hc), np.arange(wc), indexing="ij")
    cidx = np.stack([np.clip(yy + 1 + (xx > wc // 2), 0, hc - 1) * wc + np.clip(xx - 2 + yy // 5, 0, wc - 1),      # smooth with steps
                     np.where(r.random((hc, wc)) < 0.15, r.integers(0, hc * wc, (hc, wc)), (hc - 1 - yy) * wc + xx)]).reshape(B, hc * wc).astype(np.int64)
    tp = ops.window_warp_idx(T(cidx), hc, wc, 5)
    tp[0, 5, 7, 1] += 1                                   # one irregular list: that quad (and its neighbour) run as single sub-items
    tp[1, 3, :, 0] = tp[1, 3, :, 0].flip(0)
    wi = ops.WindowIndex(tp.contiguous(), (h, w), (h, w), 1)
    fq = 2.0 * r.standard_normal((B, h * w, C), dtype=np.float32)
    fk = fq + 0.7 * r.standard_normal((B, h * w, C), dtype=np.float32)
    mq = mk = None
    if masks:
        mq = (r.random((B, h * w)) > 0.2).astype(np.uint8)
        mk = (r.random((B, h * w)) > 0.2).astype(np.uint8)
        mq[0, :40] = 0
    kw = dict(mask_q=None if mq is None else T(mq), mask_k=None if mk is None else T(mk), recip=True, want_conf=conf)
    monkeypatch.setenv("CASMTR_WINDOW_KERNEL", "pair")
    d = ops.window_match(T(fq), T(fk), wi, 1.0, **kw)
    monkeypatch.setenv("CASMTR_WINDOW_KERNEL", "dma")
    e = ops.window_match(T(fq), T(fk), wi, 1.0, **kw)
    assert torch.equal(d["next_idx"], e["next_idx"]) and torch.equal(d["next_conf"], e["next_conf"])
    if conf:
        assert torch.equal(d["conf_matrix"], e["conf_matrix"])
    else:
        assert d["conf_matrix"] is None
    o = oracle.window_match(fq, fk, N(wi.materialize()), 1.0, mq, mk, recip=True)
    assert np.array_equal(N(d["next_idx"]), o["next_idx"]), "argmax must be bit-exact vs the oracle"
    assert_close(N(d["next_conf"]), o["next_conf"], SOFTMAX_TOL, "next_conf")


# ---------------------------------------------------------------------------------------------------------------------
# token-major element kernels of the calling blocks (glue.hip)
@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,Cc", [(2, 13, 11, 64), (1, 24, 32, 256), (2, 7, 8, 1024), (1, 3, 1, 8)])
@pytest.mark.parametrize("flags", [(False, False, False), (True, True, False), (False, False, True), (True, True, True)])
def test_dwconv3x3_tokens_vs_oracle_and_torch(B, H, W, Cc, flags):
    import oracle
    from casmtr_amd import ops
    r = np.random.RandomState(B * 1000 + H * 10 + Cc)
    x = r.standard_normal((B, H * W, Cc)).astype(np.float32)
    w = (r.standard_normal((Cc, 1, 3, 3)) / 3).astype(np.float32)
    b = (0.1 * r.standard_normal(Cc)).astype(np.float32)
    pre, post, add = flags
    want = oracle.dwconv3x3_tokens(x, w, b, H, W, pre, post, add)
    xt, wt, bt = (torch.from_numpy(a).cuda() for a in (x, w, b))
    got = ops.dwconv3x3_tokens(xt, wt, bt, H, W, pre, post, add)
    if post:   # erff: device libm vs glibc differ in the last ulps
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-6, atol=2e-7)
    else:
        assert np.array_equal(got.cpu().numpy(), want)
    # torch's own formulation (NCHW conv2d): tolerance only, its accumulation order is the library's
    f = xt.transpose(1, 2).reshape(B, Cc, H, W)
    f = torch.relu(f) if pre else f
    ref = torch.nn.functional.conv2d(f, wt, bt, padding=1, groups=Cc)
    ref = torch.nn.functional.gelu(ref) if post else ref
    ref = ref.flatten(2).transpose(1, 2) + (xt if add else 0)
    assert float((got - ref).abs().max()) < 1e-4
    # no bias
    got0 = ops.dwconv3x3_tokens(xt, wt, None, H, W, pre, post, add)
    want0 = oracle.dwconv3x3_tokens(x, w, None, H, W, pre, post, add)
    np.testing.assert_allclose(got0.cpu().numpy(), want0, rtol=2e-6, atol=2e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,Cc", [(37, 64), (1000, 128), (513, 256), (9, 1024), (5, 8), (64, 520)])
@pytest.mark.parametrize("with_res", [False, True])
def test_layer_norm_vs_oracle_and_torch(rows, Cc, with_res):
    import oracle
    from casmtr_amd import ops
    r = np.random.RandomState(rows + Cc)
    x = (r.standard_normal((rows, Cc)) * 2 + 0.5).astype(np.float32)
    g, b = (1 + 0.1 * r.standard_normal(Cc)).astype(np.float32), (0.1 * r.standard_normal(Cc)).astype(np.float32)
    res = r.standard_normal((rows, Cc)).astype(np.float32) if with_res else None
    want = oracle.layer_norm(x, g, b, 1e-5, res)
    xt, gt, bt = (torch.from_numpy(a).cuda() for a in (x, g, b))
    rt = torch.from_numpy(res).cuda() if with_res else None
    got = ops.layer_norm(xt, gt, bt, 1e-5, rt)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-5, atol=2e-6)
    ref = torch.nn.functional.layer_norm(xt, (Cc,), gt, bt, 1e-5) + (rt if with_res else 0)
    assert float((got - ref).abs().max()) < 1e-5


def _window_attn_torch(qkv, H, W, nh, ws, scale):
    """the reference's padded + masked formulation (GroupAttention.forward_mask), restated on torch ops"""
    import torch.nn.functional as F
    B, N, C3 = qkv.shape
    Cc = C3 // 3
    pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws
    x = F.pad(qkv.view(B, H, W, C3), (0, 0, 0, pr, 0, pb))
    Hp, Wp = H + pb, W + pr
    gh, gw = Hp // ws, Wp // ws
    pad = torch.zeros((1, Hp, Wp), device=qkv.device)
    if pb:
        pad[:, -pb:, :] = 1
    if pr:
        pad[:, :, -pr:] = 1
    pad = pad.reshape(1, gh, ws, gw, ws).transpose(2, 3).reshape(1, gh * gw, ws * ws)
    bias = pad.unsqueeze(2) - pad.unsqueeze(3)
    bias = torch.where(bias != 0, torch.full_like(bias, -1000.0), torch.zeros_like(bias))
    t = x.reshape(B, gh, ws, gw, ws, 3, nh, Cc // nh).transpose(2, 3).reshape(B, gh * gw, ws * ws, 3, nh, Cc // nh).permute(3, 0, 1, 4, 2, 5)
    att = ((t[0] @ t[1].transpose(-2, -1)) * scale + bias.unsqueeze(2)).softmax(dim=-1)
    out = (att @ t[2]).transpose(2, 3).reshape(B, gh, gw, ws, ws, Cc).transpose(2, 3).reshape(B, Hp, Wp, Cc)
    return out[:, :H, :W, :].reshape(B, N, Cc)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,nh", [(2, 14, 21, 4), (1, 24, 32, 8), (2, 13, 9, 4), (1, 5, 3, 2), (1, 52, 52, 4)])
def test_window_attn_vs_oracle_and_torch(B, H, W, nh):
    import oracle
    from casmtr_amd import ops
    r = np.random.RandomState(H * 100 + W)
    Cc = nh * 32
    qkv = r.standard_normal((B, H * W, 3 * Cc)).astype(np.float32)
    scale = 32 ** -0.5
    want = oracle.window_attn(qkv, H, W, nh, 7, scale)
    qt = torch.from_numpy(qkv).cuda()
    got = ops.window_attn(qt, H, W, nh, 7, scale)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-5, atol=2e-6)
    ref = _window_attn_torch(qt, H, W, nh, 7, scale)
    assert float((got - ref).abs().max()) < 2e-5
    with pytest.raises(RuntimeError):
        ops.window_attn(qt, H, W, nh, 5, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,nh", [(2, 14, 21, 4), (1, 13, 9, 4), (1, 30, 40, 2), (1, 5, 3, 8)])
def test_pola_attn_vs_reference_formulation(B, H, W, nh):
    """casmtr_pola_attn_fwd against the reference's formulation restated on torch ops: pad to a multiple of 7, one more window of
    zeros all round, unfold the 21 x 21 neighbourhoods, project them WITH bias, add the relative position bias, softmax"""
    import torch.nn.functional as F
    from casmtr_amd import ops
    g = torch.Generator(device="cuda").manual_seed(H * 100 + W)
    Cc, ws, n = nh * 32, 7, 3
    x = torch.randn(B, H * W, Cc, device="cuda", generator=g)
    Wq, Wk, Wv = (torch.randn(Cc, Cc, device="cuda", generator=g) / Cc ** 0.5 for _ in range(3))
    bq, bk, bv = (0.3 * torch.randn(Cc, device="cuda", generator=g) for _ in range(3))
    table = 0.5 * torch.randn((4 * ws - 1) ** 2, nh, device="cuda", generator=g)
    scale = 32 ** -0.5
    got = ops.pola_attn(F.linear(x, Wq, bq).contiguous(), F.linear(x, Wk).contiguous(), F.linear(x, Wv).contiguous(), table, H, W, nh, ws,
                        scale) + bv
    # ---- reference formulation
    pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws
    xp = F.pad(x.view(B, H, W, Cc), (0, 0, 0, pr, 0, pb))
    Hp, Wp = H + pb, W + pr
    gh, gw = Hp // ws, Wp // ws
    xq = xp.view(B, gh, ws, gw, ws, Cc).permute(0, 1, 3, 2, 4, 5).reshape(B * gh * gw, ws * ws, Cc)
    kv = F.unfold(F.pad(xp, (0, 0, ws, ws, ws, ws)).permute(0, 3, 1, 2), n * ws, stride=ws)
    kv = kv.permute(0, 2, 1).reshape(B * gh * gw, Cc, (n * ws) ** 2).permute(0, 2, 1)
    hd = lambda t: t.view(t.shape[0], t.shape[1], nh, 32).transpose(1, 2)
    q, k, v = hd(F.linear(xq, Wq, bq)) * scale, hd(F.linear(kv, Wk, bk)), hd(F.linear(kv, Wv, bv))
    qq = torch.arange(ws, device="cuda"); kk = torch.arange(n * ws, device="cuda")
    qy, qx = torch.meshgrid(qq, qq, indexing="ij"); ky, kx = torch.meshgrid(kk, kk, indexing="ij")
    idx = (qy.reshape(-1, 1) - ky.reshape(1, -1) + n * ws - 1) * (4 * ws - 1) + (qx.reshape(-1, 1) - kx.reshape(1, -1) + n * ws - 1)
    bias = table[idx.view(-1)].view(ws * ws, -1, nh).permute(2, 0, 1)
    att = (q @ k.transpose(-2, -1) + bias.unsqueeze(0)).softmax(-1)
    ref = (att @ v).transpose(1, 2).reshape(B, gh, gw, ws, ws, Cc).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, Cc)[:, :H, :W].reshape(B, H * W, Cc)
    assert float((got - ref).abs().max()) < 5e-5
