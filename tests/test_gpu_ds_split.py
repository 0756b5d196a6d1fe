"""Split dual-softmax (casmtr_dual_softmax_split_fwd, csrc/ds_split.hip): the similarity matrix comes from the f16 matrix pipe, the
indices must still be the oracle's.  CoarseMatching.forward, src/model/functions/coarse_matching.py:59-89.

  * error bound: |sim_split - sim_exact| <= 2^-15 |a_i||b_j|/(C T) is what the candidate margin assumes; measured here with a
    factor 8 to spare, on random, badly scaled and sparse rows;
  * near ties: columns that differ from another column by a few ulp in one channel -> several candidates per row, the exact chain
    decides; the result must be the oracle's first maximum;
  * overflow: more than DS_CAND_CAP equal rows -> the device-side fallback to the exact passes;
  * masks: fully masked rows / columns answer index 0 (first of the equal -1e9 entries).
"""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
N = lambda t: t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
    from casmtr_amd import ops as o
    return o


def _features(kind, r, B, L, C):
    f = r.standard_normal((B, L, C), dtype=np.float32)
    if kind == "scaled":      # rows spanning 7 orders of magnitude, a few huge / tiny channels inside a row
        f *= np.exp(r.uniform(-10, 6, (B, L, 1))).astype(np.float32)
        f[:, :, 3] *= 300.0
        f[:, :, 7] *= 1e-4
    elif kind == "sparse":    # most channels exactly zero
        f *= (r.random((B, L, C)) < 0.1)
    return np.ascontiguousarray(f, dtype=np.float32)


@pytest.mark.parametrize("kind", ["randn", "scaled", "sparse"])
def test_split_error_bound(ops, kind):
    r = np.random.default_rng(11)
    B, h, w, C, Tm = 2, 24, 20, 256, 0.1      # 480 tokens: 4 row blocks, the last one partial
    f0, f1 = _features(kind, r, B, h * w, C), _features(kind, r, B, h * w, C)
    ex = ops.dual_softmax(T(f0), T(f1), (h, w), (h, w), Tm, 0.2, want_conf=False, gemm="exact", want_sim=True)
    sp = ops.dual_softmax(T(f0), T(f1), (h, w), (h, w), Tm, 0.2, want_conf=False, gemm="split", want_sim=True)
    na = np.linalg.norm(f0.astype(np.float64) / 16.0, axis=2)
    nb = np.linalg.norm(f1.astype(np.float64) / 16.0, axis=2)
    bound = 2.0 ** -15 * na[:, :, None] * nb[:, None, :] / Tm
    err = np.abs(N(sp["sim"]).astype(np.float64) - N(ex["sim"]).astype(np.float64))
    ratio = (err / np.maximum(bound, 1e-300)).max()
    assert ratio < 0.125, f"{kind}: split error is {ratio:.3f} of the candidate margin's bound"
    assert torch.equal(sp["next_idx_c01"], ex["next_idx_c01"]) and torch.equal(sp["next_idx_c10"], ex["next_idx_c10"])
    if kind != "scaled":   # logits of 1e6 and more: their fp32 rounding alone moves the probabilities, in either path
        assert np.abs(N(sp["next_conf_c01"]) - N(ex["next_conf_c01"])).max() < 1e-5


def _near_duplicates(r, f, ndup, copies):
    """rows of f[0] copied `copies` times with one channel moved by 0..2 ulp (equal, or separated far below the split error)"""
    L, C = f.shape[1:]
    for _ in range(ndup):
        src = int(r.integers(0, L))
        for dst in r.choice(L, copies, replace=False):
            f[0, dst] = f[0, src]
            c = int(r.integers(0, C))
            for _ in range(int(r.integers(0, 3))):
                f[0, dst, c] = np.nextafter(f[0, dst, c], np.float32(np.inf))
    return f


@pytest.mark.parametrize("recip", [False, True])
@pytest.mark.parametrize("copies,masks", [(2, False), (5, True), (12, False)])
def test_split_near_ties_and_overflow(ops, copies, masks, recip):
    """copies = 2, 5: candidate lists of 3-6 entries, decided by the exact chain; 12: more than DS_CAND_CAP -> exact fallback"""
    r = np.random.default_rng(100 + copies)
    h, w, C = 16, 18, 256
    L = h * w
    f0 = _near_duplicates(r, r.standard_normal((1, L, C), dtype=np.float32), 20, copies)
    f1 = _near_duplicates(r, r.standard_normal((1, L, C), dtype=np.float32), 20, copies)
    m0 = m1 = valid = None
    if masks:
        mm0, mm1 = np.ones((1, h, w), bool), np.ones((1, h, w), bool)
        mm0[:, 13:], mm0[:, :, 15:] = False, False
        mm1[:, 12:], mm1[:, :, 16:] = False, False
        m0, m1 = mm0.reshape(1, -1), mm1.reshape(1, -1)
        valid = np.array([[13, 15, 12, 16]], np.int32)
    o = oracle.dual_softmax(f0, f1, (h, w), (h, w), 0.1, 0.2, mask0=m0, mask1=m1, valid_hw=valid, recip=recip)
    tm = lambda m: None if m is None else T(m)
    d = ops.dual_softmax(T(f0), T(f1), (h, w), (h, w), 0.1, 0.2, mask0=tm(m0), mask1=tm(m1), valid_hw=tm(valid), recip=recip,
                         want_conf=False, gemm="split")
    assert np.array_equal(N(d["next_idx_c01"]), o["next_idx_c01"])
    assert np.array_equal(N(d["next_idx_c10"]), o["next_idx_c10"])
    assert np.abs(N(d["next_conf_c01"]) - o["next_conf_c01"]).max() < 1e-5
    assert np.abs(N(d["next_conf_c10"]) - o["next_conf_c10"]).max() < 1e-5
    n = int(d["n"].item())
    assert n == len(o["i_ids"]) and np.array_equal(N(d["i_ids"][:n]), o["i_ids"]) and np.array_equal(N(d["j_ids"][:n]), o["j_ids"])


def test_split_degenerate_inputs(ops):
    """all-zero features: every entry ties -> overflow -> exact passes; index 0 everywhere, uniform probabilities"""
    h, w, C = 12, 12, 64
    z = torch.zeros((1, h * w, C), device=DEV)
    d = ops.dual_softmax(z, z, (h, w), (h, w), 0.1, 0.2, want_conf=True, gemm="split")
    assert int(d["next_idx_c01"].abs().max()) == 0 and int(d["next_idx_c10"].abs().max()) == 0
    assert torch.allclose(d["next_conf_c01"], torch.full_like(d["next_conf_c01"], 1.0 / (h * w)))
    assert torch.allclose(d["conf_matrix"], torch.full_like(d["conf_matrix"], 1.0 / (h * w) ** 2))


def test_split_full_size_vs_exact(ops):
    """832x832 grid (104^2 tokens, C = 256), two pairs, 20 % padding masks on the second call: the two GEMM paths agree on every
    index and to 1e-6 on the probabilities; match lists identical."""
    g = torch.Generator(device="cpu").manual_seed(3)
    B, h, C = 2, 104, 256
    f0 = torch.randn((B, h * h, C), generator=g).to(DEV)
    f1 = torch.randn((B, h * h, C), generator=g).to(DEV)
    for masked in (False, True):
        m0 = m1 = valid = None
        if masked:
            m = torch.ones((B, h, h), dtype=torch.bool)
            m[:, 83:], m[:, :, 90:] = False, False
            m0 = m1 = m.reshape(B, -1).to(DEV)
            valid = torch.tensor([[83, 90, 83, 90]] * B, dtype=torch.int32, device=DEV)
        ex = ops.dual_softmax(f0, f1, (h, h), (h, h), 0.1, 0.2, mask0=m0, mask1=m1, valid_hw=valid, want_conf=False, gemm="exact")
        sp = ops.dual_softmax(f0, f1, (h, h), (h, h), 0.1, 0.2, mask0=m0, mask1=m1, valid_hw=valid, want_conf=False, gemm="split")
        assert torch.equal(sp["next_idx_c01"], ex["next_idx_c01"]) and torch.equal(sp["next_idx_c10"], ex["next_idx_c10"])
        assert float((sp["next_conf_c01"] - ex["next_conf_c01"]).abs().max()) < 1e-6
        assert float((sp["next_conf_c10"] - ex["next_conf_c10"]).abs().max()) < 1e-6
        n = int(sp["n"].item())
        assert n == int(ex["n"].item())
        assert torch.equal(sp["i_ids"][:n], ex["i_ids"][:n]) and torch.equal(sp["j_ids"][:n], ex["j_ids"][:n])
        assert float((sp["mconf"][:n] - ex["mconf"][:n]).abs().max()) < 1e-6 if n else True


@pytest.mark.parametrize("hw0,hw1", [((40, 36), (28, 40)), ((16, 16), (24, 24)), ((52, 52), (52, 52))])
@pytest.mark.parametrize("masked", [False, True])
def test_split_with_and_without_the_stored_matrix(ops, hw0, hw1, masked):
    """Round 6: without conf_matrix the split path no longer writes the similarity matrix; its pass 2 recomputes the flagged (row, 128-column)
    / (column, 128-row) segments from the operand images with the GEMM's own MFMA sequence (csrc/ds_split.hip: ds_flagged_launch).  Every
    output must equal, bit for bit, what the same call produces when the matrix is stored as well (want_sim: the GEMM's store epilogue)
    and what the dense pass 2 (want_conf: it streams the stored matrix) decides.  1440 x 1120 (ragged tiles both ways), 256 x 576 (one
    row block), 2704 x 2704; with and without padding masks."""
    g = torch.Generator(device="cpu").manual_seed(5)
    B, C = 2, 256
    L, S = hw0[0] * hw0[1], hw1[0] * hw1[1]
    f0 = torch.randn((B, L, C), generator=g)
    f1 = torch.randn((B, S, C), generator=g)
    n_m = min(L, S) // 2   # half of the rows have a true match somewhere (strong maxima), the rest compete at noise level
    for b in range(B):
        src, dst = torch.randperm(L, generator=g)[:n_m], torch.randperm(S, generator=g)[:n_m]
        f1[b, dst] = f0[b, src] + 0.4 * torch.randn((n_m, C), generator=g)
    f0, f1 = f0.to(DEV), f1.to(DEV)
    m0 = m1 = valid = None
    if masked:
        a = torch.ones((B,) + hw0, dtype=torch.bool)
        c = torch.ones((B,) + hw1, dtype=torch.bool)
        a[:, hw0[0] - 5:], a[:, :, hw0[1] - 3:] = False, False
        c[:, hw1[0] - 2:], c[:, :, hw1[1] - 7:] = False, False
        m0, m1 = a.reshape(B, -1).to(DEV), c.reshape(B, -1).to(DEV)
        valid = torch.tensor([[hw0[0] - 5, hw0[1] - 3, hw1[0] - 2, hw1[1] - 7]] * B, dtype=torch.int32, device=DEV)
    run = lambda **kw: ops.dual_softmax(f0, f1, hw0, hw1, 0.1, 0.2, mask0=m0, mask1=m1, valid_hw=valid, gemm="split", **kw)
    ref = run(want_conf=True)                      # dense pass 2 over the stored matrix
    n = int(ref["n"].item())
    assert n > 20
    for kw in (dict(want_conf=False), dict(want_conf=False, want_sim=True)):
        out = run(**kw)
        assert (out["sim"] is not None) == bool(kw.get("want_sim"))
        assert int(out["n"].item()) == n
        for k in ("next_idx_c01", "next_idx_c10", "next_conf_c01", "next_conf_c10"):
            assert torch.equal(out[k], ref[k]), (kw, k)
        for k in ("i_ids", "j_ids", "b_ids", "mconf"):
            assert torch.equal(out[k][:n], ref[k][:n]), (kw, k)
    # the exact (all-fp32) path likewise: no matrix (recomputed fp32-chain segments) == matrix stored == conf_matrix written, bit for bit
    exs = [ops.dual_softmax(f0, f1, hw0, hw1, 0.1, 0.2, mask0=m0, mask1=m1, valid_hw=valid, gemm="exact", **kw)
           for kw in (dict(want_conf=True), dict(want_conf=False), dict(want_conf=False, want_sim=True))]
    assert exs[1]["sim"] is None and exs[2]["sim"] is not None
    for ex in exs:
        assert torch.equal(ex["next_idx_c01"], ref["next_idx_c01"]) and torch.equal(ex["next_idx_c10"], ref["next_idx_c10"])
        assert int(ex["n"].item()) == n and torch.equal(ex["i_ids"][:n], ref["i_ids"][:n]) and torch.equal(ex["j_ids"][:n], ref["j_ids"][:n])
    for ex in exs[1:]:
        for k in ("next_conf_c01", "next_conf_c10"):
            assert torch.equal(ex[k], exs[0][k]), k
        assert torch.equal(ex["mconf"][:n], exs[0]["mconf"][:n]) and torch.equal(ex["b_ids"][:n], exs[0]["b_ids"][:n])


def test_split_low_threshold_uses_dense_pass(ops):
    """thr < 1e-3: the sparse pass 2 (whose segment test needs log(thr * rsum)) hands over to the dense one; same lists as the exact path"""
    g = torch.Generator(device="cpu").manual_seed(9)
    B, h, w, C = 2, 20, 24, 256
    f0 = torch.randn((B, h * w, C), generator=g).to(DEV)
    f1 = (f0[:, torch.randperm(h * w, generator=g)] + 0.3 * torch.randn((B, h * w, C), generator=g).to(DEV)).contiguous()
    for thr in (0.0, 5e-4):
        ex = ops.dual_softmax(f0, f1, (h, w), (h, w), 0.1, thr, want_conf=False, gemm="exact")
        sp = ops.dual_softmax(f0, f1, (h, w), (h, w), 0.1, thr, want_conf=False, gemm="split")
        n = int(sp["n"].item())
        assert n == int(ex["n"].item()) and n > 100
        assert torch.equal(sp["i_ids"][:n], ex["i_ids"][:n]) and torch.equal(sp["j_ids"][:n], ex["j_ids"][:n])
        assert torch.equal(sp["next_idx_c01"], ex["next_idx_c01"]) and torch.equal(sp["next_idx_c10"], ex["next_idx_c10"])


def _lists(d):
    n = int(d["n"].item())
    return n, d["b_ids"][:n], d["i_ids"][:n], d["j_ids"][:n], d["mconf"][:n]


def _assert_same_lists(sp, ex, what):
    ns, bs, is_, js, cs = _lists(sp)
    ne, be, ie, je, ce = _lists(ex)
    assert ns == ne, f"{what}: {ns} matches on the split path, {ne} on the exact path"
    assert torch.equal(bs, be) and torch.equal(is_, ie) and torch.equal(js, je), f"{what}: match lists differ"
    return cs, ce


@pytest.mark.parametrize("want_conf", [False, True])   # sparse pass 2 | dense pass 2 (conf_matrix written)
def test_split_match_list_exact_by_construction(ops, want_conf):
    """VERDICT r04 item 3.  The split path's (b, i, j) list must EQUAL the exact path's on every input, not merely in practice.
    Decision boundaries are planted exactly where the approximate confidences cannot be trusted:
      * thr set to the exact confidence of a match, and to its float neighbours -- `conf > thr` flips between neighbouring thresholds,
        the split path's conf of that entry is off by ~1e-6 relative, i.e. thousands of ulps;
      * bit-identical and one-ulp-apart columns / rows -- row and column maxima tie exactly or within an ulp, so the first-index rule
        and the mutual-maximum-by-value test are decided on the last bit.
    Entries near a boundary are re-decided from exact logits and exact statistics (ds_split.hip: ds_xdecide_launch); their mconf must
    then be the exact path's value bit for bit."""
    g = torch.Generator(device="cpu").manual_seed(21)
    B, h, w, C = 2, 24, 28, 256
    L = h * w
    f0 = torch.randn((B, L, C), generator=g)
    perm = torch.stack([torch.randperm(L, generator=g) for _ in range(B)])
    f1 = torch.stack([f0[b][perm[b]] for b in range(B)]) + 0.45 * torch.randn((B, L, C), generator=g)
    # near-duplicate columns / rows: exact ties and one-ulp differences between competing maxima
    for b in range(B):
        for k in range(12):
            src, dst = int(torch.randint(0, L, (1,), generator=g)), int(torch.randint(0, L, (1,), generator=g))
            f1[b, dst] = f1[b, src]
            if k % 3:
                c = int(torch.randint(0, C, (1,), generator=g))
                f1[b, dst, c] = torch.nextafter(f1[b, dst, c], torch.tensor(float("inf") if k % 3 == 1 else -float("inf")))
            src, dst = int(torch.randint(0, L, (1,), generator=g)), int(torch.randint(0, L, (1,), generator=g))
            f0[b, dst] = f0[b, src]
    f0, f1 = f0.to(DEV).contiguous(), f1.to(DEV).contiguous()
    run = lambda gemm, thr: ops.dual_softmax(f0, f1, (h, w), (h, w), 0.1, thr, want_conf=want_conf, gemm=gemm)
    ex = run("exact", 0.2)
    n, _, _, _, mc = _lists(ex)
    assert n > 200
    _assert_same_lists(run("split", 0.2), ex, "thr = 0.2")
    # thresholds ON exact confidences: the five smallest confidences above 0.2, a median one and the largest one
    srt = torch.sort(mc).values
    picks = [float(srt[k]) for k in (0, 1, 2, 3, 4, n // 2, n - 1)]
    checked = 0
    for c in picks:
        for thr in (float(np.nextafter(np.float32(c), np.float32(0))), c, float(np.nextafter(np.float32(c), np.float32(1)))):
            e2, s2 = run("exact", thr), run("split", thr)
            cs, ce = _assert_same_lists(s2, e2, f"thr = {thr!r} (at an exact confidence {c!r})")
            # entries within the band of thr were re-decided: their mconf is the exact value bit for bit
            near = (ce - thr).abs() <= 1e-5 * thr
            assert torch.equal(cs[near], ce[near]), "re-decided entries carry the exact confidence"
            assert float((cs - ce).abs().max()) < 1e-5 if cs.numel() else True
            checked += int(near.sum())
    assert checked >= len(picks), "the planted thresholds actually had entries sitting on them"
    # indices stay the oracle's as before
    o = oracle.dual_softmax(f0.cpu().numpy(), f1.cpu().numpy(), (h, w), (h, w), 0.1, 0.2, recip=True)
    sp = run("split", 0.2)
    assert np.array_equal(N(sp["next_idx_c01"]), o["next_idx_c01"]) and np.array_equal(N(sp["next_idx_c10"]), o["next_idx_c10"])


def test_split_borderline_overflow_falls_back_to_exact(ops):
    """hundreds of rows with the same top confidence pattern (every row duplicated many times): more borderline entries than the
    list holds -> the device-side flag sends the call through the exact passes, lists still equal the exact path's"""
    g = torch.Generator(device="cpu").manual_seed(5)
    h, w, C = 20, 20, 64
    L = h * w
    base = torch.randn((1, 40, C), generator=g)
    f0 = base[:, torch.arange(L) % 40].contiguous()                    # 10 copies of each of 40 rows
    f1 = (base[:, torch.arange(L) % 40] * 1.0).contiguous()
    f0, f1 = f0.to(DEV), f1.to(DEV)
    for thr in (0.0005, 0.002):
        ex = ops.dual_softmax(f0, f1, (h, w), (h, w), 0.1, thr, want_conf=False, gemm="exact")
        sp = ops.dual_softmax(f0, f1, (h, w), (h, w), 0.1, thr, want_conf=False, gemm="split")
        _assert_same_lists(sp, ex, f"duplicated rows, thr = {thr}")


@pytest.mark.parametrize("want_conf", [False, True])
def test_split_large_norm_features_fall_back_to_exact(ops, want_conf):
    """ADVICE r05 (ds_conf_band has no cap).  Unnormalised, large-norm features: |a||b| / (C T) in the hundreds makes the confidence
    band of the split path wider than the 0.1 margin the borderline lists assume (they drop entries below 0.9 thr), so a runner-up
    that the exact path accepts could go unlisted.  Such a pair must take the exact passes (ds_xnear_kernel raises the fallback
    flag when band > DS_BAND_MAX): lists, indices and confidences are then the exact path's."""
    g = torch.Generator(device="cpu").manual_seed(33)
    B, h, w, C = 2, 20, 24, 256
    L = h * w
    f0 = 6.0 * torch.randn((B, L, C), generator=g)            # |a| / sqrt(C) ~ 6: namax * nbmax ~ 40 > 19 (band > 0.04 at T = 0.1)
    perm = torch.stack([torch.randperm(L, generator=g) for _ in range(B)])
    f1 = torch.stack([f0[b][perm[b]] for b in range(B)]) + 1.5 * torch.randn((B, L, C), generator=g)
    f0[1] *= 0.1                                               # the second pair stays inside the band's assumptions (per-pair bound)
    f1[1] *= 0.1
    f0, f1 = f0.to(DEV).contiguous(), f1.to(DEV).contiguous()
    for thr in (0.2, 0.02):
        ex = ops.dual_softmax(f0, f1, (h, w), (h, w), 0.1, thr, want_conf=want_conf, gemm="exact")
        sp = ops.dual_softmax(f0, f1, (h, w), (h, w), 0.1, thr, want_conf=want_conf, gemm="split")
        _assert_same_lists(sp, ex, f"large-norm features, thr = {thr}")
        assert torch.equal(sp["next_idx_c01"], ex["next_idx_c01"]) and torch.equal(sp["next_idx_c10"], ex["next_idx_c10"])
        assert torch.equal(sp["next_conf_c01"], ex["next_conf_c01"]), "the exact passes ran: confidences are the exact path's bit for bit"


@pytest.mark.parametrize("C", [64, 128, 192, 256])
def test_fused_prepass_equals_three_pass(ops, C, monkeypatch):
    """ds_prep_kernel (one launch: exponents, factors, norms, batch maxima and both tile images) against ds_rownorm_kernel +
    ds_nmax_kernel + ds_split_kernel: the similarity matrix the GEMM computes from the images, and everything downstream, bit for bit;
    ragged row blocks, padding masks, badly scaled and all-zero rows"""
    g = torch.Generator(device="cpu").manual_seed(C)
    B, hw0, hw1 = 2, (20, 23), (17, 30)
    L, S = hw0[0] * hw0[1], hw1[0] * hw1[1]
    f0 = torch.randn((B, L, C), generator=g) * torch.exp(torch.empty((B, L, 1)).uniform_(-8, 8, generator=g))
    f1 = torch.randn((B, S, C), generator=g)
    f1[:, :200] = f0[:, :200] + 0.3 * torch.randn((B, 200, C), generator=g)
    f0[0, 5] = 0.0
    f1[1, 7] = 0.0
    m0 = torch.ones((B, L), dtype=torch.bool)
    m1 = torch.ones((B, S), dtype=torch.bool)
    m0[:, L - 40:], m1[:, S - 25:] = False, False
    f0, f1, m0, m1 = f0.to(DEV), f1.to(DEV), m0.to(DEV), m1.to(DEV)
    outs = {}
    for mode in ("3", "1"):
        monkeypatch.setenv("CASMTR_DS_PREP", mode)
        outs[mode] = [ops.dual_softmax(f0, f1, hw0, hw1, 0.1, 0.2, mask0=mm0, mask1=mm1, want_conf=False, want_sim=True, gemm="split")
                      for mm0, mm1 in ((None, None), (m0, m1))]
    for a, b in zip(outs["3"], outs["1"]):
        for k in ("sim", "next_idx_c01", "next_idx_c10", "next_conf_c01", "next_conf_c10"):
            assert torch.equal(a[k], b[k]), k
        n = int(a["n"].item())
        assert int(b["n"].item()) == n and torch.equal(a["i_ids"][:n], b["i_ids"][:n]) and torch.equal(a["mconf"][:n], b["mconf"][:n])
