"""casmtr_amd.model.CasMTR4c against the reference's CasMTR (cascade_model_stage3.py) evaluated on CPU by
tests/golden/gen_golden_model.py -- same deterministic weights (golden_inputs.model_state), london_bridge pair at 256x192.

CPU tests: checkpoint layout (every state-dict key / shape of the reference), the torch-only pieces (backbone, fine stage).
GPU tests: every stage on the reference's stage inputs (bit-identical where the fixture stores them), then the whole forward."""
import json
import os

import numpy as np
import pytest
import torch

from golden_inputs import model_state

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "model_london_bridge.npz")


@pytest.fixture(scope="module")
def fx():
    z = np.load(FIX)
    return {k: z[k] for k in z.files}


def _config(fx):
    from casmtr_amd.model import outdoor_4c_config
    c = outdoor_4c_config()
    thr = fx["thresholds"]
    c["match_coarse"]["thr"] = float(thr[0])
    c["match_cascade"].update(test_thr=float(thr[1]), pre_thr=[float(thr[2])], double_check=bool(thr[3]))
    return c


def _load_deterministic(m):
    sd = m.state_dict()
    for k, v in model_state({k: tuple(v.shape) for k, v in sd.items()}).items():
        if sd[k].dtype == torch.float32:   # integer buffers (window offsets, relative_position_index) keep their constructor values
            sd[k] = torch.from_numpy(v)
    m.load_state_dict(sd)
    return m


def _model(fx, device):
    from casmtr_amd.model import CasMTR4c
    return _load_deterministic(CasMTR4c(_config(fx)).eval()).to(device)


def _images(fx, device):
    return [torch.from_numpy(fx[k]).to(device).float() / 255.0 for k in ("image0", "image1")]


def _close(a, b, tol, what, frac=1.0):
    a, b = a.detach().float().cpu(), torch.as_tensor(np.asarray(b, dtype=np.float32))
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs() / (1.0 + b.abs())
    ok = float((err <= tol).float().mean())
    assert ok >= frac, f"{what}: {100 * ok:.3f}% within {tol} (max err {float(err.max()):.3e})"


def test_state_dict_layout_matches_reference():
    from casmtr_amd.model import CasMTR4c
    with open(os.path.join(HERE, "golden", "model_state_keys.json")) as f:
        ref = json.load(f)
    mine = {k: list(v.shape) for k, v in CasMTR4c().state_dict().items()}
    assert sorted(mine) == sorted(ref)
    assert all(mine[k] == ref[k] for k in ref)
    # 'matcher.'-prefixed checkpoints (the lightning wrapper's) load as well
    m = CasMTR4c()
    m.load_state_dict({"matcher." + k: v for k, v in m.state_dict().items()})


def test_backbone_cpu(fx):
    m = _model(fx, "cpu")
    im0, im1 = _images(fx, "cpu")
    with torch.no_grad():
        f8, f4, ff = m.backbone(torch.cat([im0, im1], 0))
    _close(f8[:, ::4, ::2, ::2], fx["bb_f8_sub"], 2e-4, "1/8 features")
    _close(f4[:, ::4, ::4, ::4], fx["bb_f4_sub"], 2e-4, "1/4 features")
    _close(ff[:, ::4, ::8, ::8], fx["bb_ff_sub"], 2e-4, "1/2 features")


def _stage4_from_fixture(fx, device):
    t = lambda k, dt=torch.int64: torch.from_numpy(fx[k]).to(device).to(dt)
    return {"b_ids": t("m4_b_ids"), "i_ids": t("m4_i_ids"), "j_ids": t("m4_j_ids"), "m_bids": t("m4_b_ids"),
            "mconf": t("m4_mconf", torch.float32), "mkpts0_c": t("m4_mkpts0_c", torch.float32), "mkpts1_c": t("m4_mkpts1_c", torch.float32)}


def _fine_stage(fx, device):
    m = _model(fx, device)
    im0, im1 = _images(fx, device)
    data = {"image0": im0, "image1": im1}
    with torch.no_grad():
        _, _, (ff0, ff1) = m.features(data)
        t4 = torch.from_numpy(fx["t4"]).to(device).float()
        data["stage_4c"] = _stage4_from_fixture(fx, device)
        m.fine_stage(ff0, ff1, t4[:1], t4[1:], data)
    assert len(fx["mkpts1_f"]) >= 50
    _close(data["mkpts0_f"], fx["mkpts0_f"], 0, "mkpts0_f")
    _close(data["expec_f"], fx["expec_f"], 1e-3, "expec_f")
    assert float((data["mkpts1_f"].cpu() - torch.from_numpy(fx["mkpts1_f"])).abs().max()) < 5e-3   # pixels


def test_fine_stage_cpu(fx):
    _fine_stage(fx, "cpu")


@pytest.mark.gpu
def test_fine_stage_gpu(fx):
    _fine_stage(fx, "cuda")


@pytest.mark.gpu
def test_backbone_gpu(fx):
    m = _model(fx, "cuda")
    im0, im1 = _images(fx, "cuda")
    with torch.no_grad():
        f8, f4, ff = m.backbone(torch.cat([im0, im1], 0))
    _close(f8[:, ::4, ::2, ::2], fx["bb_f8_sub"], 1e-3, "1/8 features")
    _close(f4[:, ::4, ::4, ::4], fx["bb_f4_sub"], 1e-3, "1/4 features")
    _close(ff[:, ::4, ::8, ::8], fx["bb_ff_sub"], 1e-3, "1/2 features")


def _sizes(fx, data):
    H, W = fx["image0"].shape[2:]
    data.update(bs=1, hw0_i=(H, W), hw1_i=(H, W), hw0_8c=(H // 8, W // 8), hw1_8c=(H // 8, W // 8), hw0_4c=(H // 4, W // 4),
                hw1_4c=(H // 4, W // 4), hw0_f=(H // 2, W // 2), hw1_f=(H // 2, W // 2))
    return data


def _agree(a, b):
    return float((torch.as_tensor(a).cpu().long() == torch.as_tensor(np.asarray(b)).long()).float().mean())


def _run_coarse_route(m, f8, fx, route):
    """coarse_stage on `route` with every QuadtreeAttention's output and per-level top-k tensors recorded, layer by layer"""
    from casmtr_amd.modules.quadtree_block import QuadtreeAttention, set_caller_layout
    set_caller_layout(m, route)
    blocks = [b for b in m.modules() if isinstance(b, QuadtreeAttention)]
    rec = []

    def hook(mod, args, out):
        lv = mod.py_att._last_levels
        rec.append({"out": out.detach().clone(),
                    "levels": [{k: l[k].clone() for k in ("topk_idx", "topk_score") if l.get(k) is not None} for l in lv]})

    hooks = [b.register_forward_hook(hook) for b in blocks]
    for b in blocks:
        b.py_att.keep_levels = True
    try:
        with torch.no_grad():
            t0, t1 = m.coarse_stage(f8[:1], f8[1:], _sizes(fx, {}))
    finally:
        for h in hooks:
            h.remove()
        for b in blocks:
            b.py_att.keep_levels = False
        set_caller_layout(m, None)
    return torch.cat([t0, t1]), rec


@pytest.mark.gpu
def test_coarse_stage_on_reference_features(fx):
    """QuadTree transformer (6 layers, three-level top-k inside each) + dual-softmax matcher on the fixture's fp16-exact 1/8
    features.  The transformer output is continuous except where a top-k near-tie flips, so a small fraction of tokens may
    differ; the matcher then runs on the reference's own tokens and must reproduce its indices.

    Two routes through the attention blocks (modules/quadtree_block.py::_quad_route).  "tokens" (the modules' default: token-major
    kernels, expf + true division in the softmax) is held to the reference: 99.5 % of the elements within 3e-3.  "quads" (opt-in:
    the hot path's quad-major kernels, hardware exponential + reciprocal; per call the same indices and values within 2e-5 of the
    other route) drifts further over the chain, and the test asserts WHY instead of a loose statistic:
      * up to the first layer whose top-k lists differ between the routes, every element of the two routes agrees to 1e-4;
      * in that layer the lists differ for a handful of (token, head) series, and each of them is a NEAR-TIE at the coarsest pyramid
        level where it differs: the entry only one route selected and the entry only the other selected carry softmax scores within
        2e-5 relative of each other (|delta logit| <= 2e-5: the 1e-6 the routes' softmaxes differ by upstream decide it);
      * everything beyond 3e-3 of the reference appears only downstream of that layer, and stays bounded (99.5 % within 1e-2)."""
    m = _model(fx, "cuda")
    f8 = torch.from_numpy(fx["f8"]).cuda().float()
    ref = fx["t8"].astype(np.float32)
    out_t, rec_t = _run_coarse_route(m, f8, fx, "tokens")
    _close(out_t, ref, 3e-3, "1/8 tokens, tokens route (default)", frac=0.995)
    out_q, rec_q = _run_coarse_route(m, f8, fx, "quads")
    assert len(rec_t) == len(rec_q) == 6
    first = None
    for li, (a, b) in enumerate(zip(rec_t, rec_q)):
        same = all(torch.equal(x["topk_idx"], y["topk_idx"]) for x, y in zip(a["levels"], b["levels"]) if "topk_idx" in x)
        if not same:
            first = li
            break
        # identical selections: the layer's outputs differ only by the two softmax implementations
        assert float((a["out"] - b["out"]).abs().max()) <= 1e-4 * max(1.0, float(a["out"].abs().max())), f"layer {li}: same top-k, different values"
    if first is None:   # no flip on this box's arithmetic: the quads route then meets the tokens route's bound outright
        _close(out_q, ref, 3e-3, "1/8 tokens, quads route (no top-k flip)", frac=0.995)
    else:
        a, b = rec_t[first], rec_q[first]
        audited = 0
        parent_diff = None   # [B, h, w, H] bool: series whose candidate list already differs (downstream of a flip one level up)
        for lv, (x, y) in enumerate(zip(a["levels"], b["levels"])):
            if "topk_idx" not in x:
                continue
            ix, iy, sx, sy = x["topk_idx"], y["topk_idx"], x["topk_score"], y["topk_score"]     # [B, L, k, H]
            Bn, L, k, H = ix.shape
            hgt = fx["image0"].shape[2] // 8 >> (2 - lv)
            wid = L // hgt
            diff = (ix.sort(dim=2).values != iy.sort(dim=2).values).any(dim=2).view(Bn, hgt, wid, H)
            fresh = diff if parent_diff is None else diff & ~parent_diff.repeat_interleave(2, 1).repeat_interleave(2, 2)
            for bb, r, c, hh in fresh.nonzero().tolist():
                t = r * wid + c
                sa = {int(i): float(s) for i, s in zip(ix[bb, t, :, hh], sx[bb, t, :, hh])}
                sb = {int(i): float(s) for i, s in zip(iy[bb, t, :, hh], sy[bb, t, :, hh])}
                only_a, only_b = sorted(set(sa) - set(sb)), sorted(set(sb) - set(sa))
                assert len(only_a) == len(only_b) >= 1
                # the entries one route dropped and the other kept sit at the selection boundary with (nearly) the same score
                va, vb = sorted(sa[i] for i in only_a), sorted(sb[i] for i in only_b)
                for p, q in zip(va, vb):
                    assert abs(p - q) <= 2e-5 * max(p, q), f"layer {first} level {lv} series {(bb, t, hh)}: not a near-tie ({p!r} vs {q!r})"
                # ... and below every entry both routes kept (it IS the boundary, not an interior disagreement)
                common = [sa[i] for i in set(sa) & set(sb)]
                assert max(va) <= min(common) * (1 + 2e-5) if common else True
                audited += 1
            parent_diff = diff
        assert 1 <= audited <= 16, f"{audited} freshly flipped (token, head) series in layer {first}"
        for li in range(first):   # upstream of the flip the routes agree everywhere
            assert float((rec_t[li]["out"] - rec_q[li]["out"]).abs().max()) <= 1e-4 * max(1.0, float(rec_t[li]["out"].abs().max()))
        assert first + 1 < 6, "the drift bound below is about layers after the flip"
        _close(out_q, ref, 1e-2, f"1/8 tokens, quads route (near-tie flip audited in layer {first})", frac=0.995)
        _close(out_q, ref, 3e-3, f"1/8 tokens, quads route (near-tie flip audited in layer {first})", frac=0.90)
    # the matcher on the reference's tokens
    t8 = torch.from_numpy(fx["t8"]).cuda().float()
    data = _sizes(fx, {})
    m.coarse_matching_8c(t8[:1].contiguous(), t8[1:].contiguous(), data, level="8c")
    s8 = data["stage_8c"]
    assert _agree(s8["next_idx_c01"], fx["m8_next_idx_c01"]) == 1.0
    assert _agree(s8["next_idx_c10"], fx["m8_next_idx_c10"]) == 1.0
    _close(s8["next_conf_c01"], fx["m8_next_conf_c01"], 1e-4, "next_conf_c01")
    assert s8["i_ids"].cpu().tolist() == fx["m8_i_ids"].astype(np.int64).tolist()
    assert s8["j_ids"].cpu().tolist() == fx["m8_j_ids"].astype(np.int64).tolist()


@pytest.mark.gpu
def test_cascade_stage_on_reference_tokens(fx):
    """UpBlock + cascade transformer (window cross-attention around the reference's 1/8 argmax, 7x7 window self-attention) on
    this model's own 1/4 backbone map and the fixture's 1/8 tokens; then the 1/4 matcher on the reference's 1/4 tokens."""
    m = _model(fx, "cuda")
    im0, im1 = _images(fx, "cuda")
    data = {"image0": im0, "image1": im1}
    t8 = torch.from_numpy(fx["t8"]).cuda().float()
    idx = lambda k: torch.from_numpy(fx[k].astype(np.int64)).cuda()
    with torch.no_grad():
        _, (f4_0, f4_1), _ = m.features(data)
        data["stage_8c"] = {"next_idx_c01": idx("m8_next_idx_c01"), "next_idx_c10": idx("m8_next_idx_c10"),
                            "next_conf_c01": torch.from_numpy(fx["m8_next_conf_c01"]).cuda(), "next_conf_c01_s": None}
        t0, t1 = m.cascade_stage(f4_0, f4_1, t8[:1], t8[1:], data)
    _close(torch.cat([t0, t1]), fx["t4"].astype(np.float32), 3e-3, "1/4 tokens")
    # the matcher on the reference's tokens and windows
    t4 = torch.from_numpy(fx["t4"]).cuda().float()
    from casmtr_amd import ops
    H4, W4 = data["hw0_4c"]
    wi = [ops.WindowIndex(ops.window_warp_idx(data["stage_8c"][k], H4 // 2, W4 // 2, 5), (H4, W4), (H4, W4), 1)
          for k in ("next_idx_c01", "next_idx_c10")]
    m.cascade_matching_4c(t4[:1].contiguous(), t4[1:].contiguous(), wi[0], wi[1], data, level="4c", pre_level="8c")
    s4 = data["stage_4c"]
    assert len(fx["m4_i_ids"]) >= 50
    assert s4["i_ids"].cpu().tolist() == fx["m4_i_ids"].astype(np.int64).tolist()
    assert s4["j_ids"].cpu().tolist() == fx["m4_j_ids"].astype(np.int64).tolist()
    _close(s4["mconf"], fx["m4_mconf"], 1e-4, "mconf")
    _close(s4["mkpts1_c"], fx["m4_mkpts1_c"], 0, "mkpts1_c")


@pytest.mark.gpu
def test_whole_forward(fx):
    """free-running forward: fp32 differences of the GPU convolutions can flip near-tied selections, so this is a statistical
    check -- the reference's matches are found again (same 1/4 cells, sub-pixel position within a quarter pixel)"""
    m = _model(fx, "cuda")
    im0, im1 = _images(fx, "cuda")
    data = m({"image0": im0, "image1": im1})
    mine = {(int(a[0]), int(a[1])): b for a, b in zip(data["mkpts0_f"].cpu().tolist(), data["mkpts1_f"].cpu())}
    ref0, ref1 = fx["mkpts0_f"], torch.from_numpy(fx["mkpts1_f"])
    hit = sum(1 for a, b in zip(ref0.tolist(), ref1) if (int(a[0]), int(a[1])) in mine
              and float((mine[(int(a[0]), int(a[1]))] - b).abs().max()) < 0.25)
    assert hit >= 0.9 * len(ref0), f"{hit} of {len(ref0)} reference matches reproduced ({len(mine)} found)"
    assert abs(len(mine) - len(ref0)) <= 0.1 * len(ref0) + 2


# ---------------------------------------------------------------------------------------------------------------------
# CasMTR-2c (cascade_model_stage4.py): the third stage at 1/2 resolution, fixture from gen_golden_model.py 2c (256x128)
FIX2 = os.path.join(HERE, "golden", "model2c_london_bridge.npz")


@pytest.fixture(scope="module")
def fx2():
    z = np.load(FIX2)
    return {k: z[k] for k in z.files}


def _model2c(fx2, device):
    from casmtr_amd.model import CasMTR2c, outdoor_2c_config
    c = outdoor_2c_config()
    thr = fx2["thresholds"]
    c["match_coarse"]["thr"] = float(thr[0])
    c["match_cascade"].update(test_thr=float(thr[1]), pre_thr=[float(thr[2])], double_check=bool(thr[3]))
    c["match_cascade_2c"].update(test_thr=float(thr[1]), pre_thr=[float(thr[2])] * 2, double_check=bool(thr[3]))
    return _load_deterministic(CasMTR2c(c).eval()).to(device)


def test_state_dict_layout_2c_matches_reference():
    from casmtr_amd.model import CasMTR2c
    with open(os.path.join(HERE, "golden", "model2c_state_keys.json")) as f:
        ref = json.load(f)
    mine = {k: list(v.shape) for k, v in CasMTR2c().state_dict().items()}
    assert sorted(mine) == sorted(ref)
    assert all(mine[k] == ref[k] for k in ref)


def _stage2_from_fixture(fx2, device):
    t = lambda k, dt=torch.int64: torch.from_numpy(fx2[k].astype(np.int64) if dt == torch.int64 else fx2[k]).to(device).to(dt)
    return {"b_ids": t("m2_b_ids"), "i_ids": t("m2_i_ids"), "j_ids": t("m2_j_ids"), "m_bids": t("m2_b_ids"),
            "mconf": t("m2_mconf", torch.float32), "mkpts0_c": t("m2_mkpts0_c", torch.float32), "mkpts1_c": t("m2_mkpts1_c", torch.float32)}


def _sizes2(fx2, data):
    H, W = fx2["image0"].shape[2:]
    data.update(bs=1, hw0_i=(H, W), hw1_i=(H, W))
    for lv, d in (("8c", 8), ("4c", 4), ("2c", 2), ("f", 2)):
        data[f"hw0_{lv}"] = data[f"hw1_{lv}"] = (H // d, W // d)
    return data


def _fine_stage_2c(fx2, device):
    m = _model2c(fx2, device)
    data = _sizes2(fx2, {})
    t2 = torch.from_numpy(fx2["t2"]).to(device).float()
    data["stage_2c"] = _stage2_from_fixture(fx2, device)
    with torch.no_grad():
        m.fine_stage(None, None, t2[:1], t2[1:], data)
    assert len(fx2["mkpts1_f"]) >= 50
    _close(data["mkpts0_f"], fx2["mkpts0_f"], 0, "mkpts0_f")
    _close(data["expec_f"], fx2["expec_f"], 1e-3, "expec_f")
    assert float((data["mkpts1_f"].cpu() - torch.from_numpy(fx2["mkpts1_f"])).abs().max()) < 5e-3   # pixels


def test_fine_stage_2c_cpu(fx2):
    _fine_stage_2c(fx2, "cpu")


@pytest.mark.gpu
def test_fine_stage_2c_gpu(fx2):
    _fine_stage_2c(fx2, "cuda")


@pytest.mark.gpu
def test_third_stage_on_reference_tokens(fx2):
    """up_block2 + the 1/2-level cascade transformer on this model's own 1/2 backbone map and the fixture's fp16-exact 1/4 tokens /
    1/4 argmax; then the 1/2 matcher (NMS, both previous levels' confidences) on the reference's 1/2 tokens."""
    m = _model2c(fx2, "cuda")
    im = [torch.from_numpy(fx2[k]).cuda().float() / 255.0 for k in ("image0", "image1")]
    data = {"image0": im[0], "image1": im[1]}
    t4 = torch.from_numpy(fx2["t4"]).cuda().float()
    idx = lambda k: torch.from_numpy(fx2[k].astype(np.int64)).cuda()
    cf = lambda k: torch.from_numpy(fx2[k]).cuda()
    with torch.no_grad():
        _, _, (ff0, ff1) = m.features(data)
        data["stage_8c"] = {"next_conf_c01": cf("m8_next_conf_c01"), "next_conf_c01_s": None}
        data["stage_4c"] = {"next_idx_c01": idx("m4_next_idx_c01"), "next_idx_c10": idx("m4_next_idx_c10"),
                            "next_conf_c01": cf("m4_next_conf_c01"), "next_conf_c01_s": None}
        t0, t1 = m.cascade_stage(ff0, ff1, t4[:1], t4[1:], data, "2c")
    _close(torch.cat([t0, t1]), fx2["t2"].astype(np.float32), 3e-3, "1/2 tokens")
    from casmtr_amd import ops
    t2 = torch.from_numpy(fx2["t2"]).cuda().float()
    H2, W2 = data["hw0_2c"]
    wi = [ops.WindowIndex(ops.window_warp_idx(data["stage_4c"][k], H2 // 2, W2 // 2, 5), (H2, W2), (H2, W2), 1)
          for k in ("next_idx_c01", "next_idx_c10")]
    m.cascade_matching_2c(t2[:1].contiguous(), t2[1:].contiguous(), wi[0], wi[1], data, level="2c", pre_level=["8c", "4c"])
    s2 = data["stage_2c"]
    assert len(fx2["m2_i_ids"]) >= 50
    assert s2["i_ids"].cpu().tolist() == fx2["m2_i_ids"].astype(np.int64).tolist()
    assert s2["j_ids"].cpu().tolist() == fx2["m2_j_ids"].astype(np.int64).tolist()
    _close(s2["mconf"], fx2["m2_mconf"], 1e-4, "mconf")


@pytest.mark.gpu
def test_whole_forward_2c(fx2):
    m = _model2c(fx2, "cuda")
    im = [torch.from_numpy(fx2[k]).cuda().float() / 255.0 for k in ("image0", "image1")]
    data = m({"image0": im[0], "image1": im[1]})
    mine = {(int(a[0]), int(a[1])): b for a, b in zip(data["mkpts0_f"].cpu().tolist(), data["mkpts1_f"].cpu())}
    ref0, ref1 = fx2["mkpts0_f"], torch.from_numpy(fx2["mkpts1_f"])
    hit = sum(1 for a, b in zip(ref0.tolist(), ref1) if (int(a[0]), int(a[1])) in mine
              and float((mine[(int(a[0]), int(a[1]))] - b).abs().max()) < 0.25)
    assert hit >= 0.9 * len(ref0), f"{hit} of {len(ref0)} reference matches reproduced ({len(mine)} found)"
    assert abs(len(mine) - len(ref0)) <= 0.1 * len(ref0) + 2


@pytest.mark.gpu
def test_reduced_precision_convolutions(fx):
    """conv_dtype=torch.float16 (the reference's test.py evaluates under fp16 autocast): only the glue's convolutions change
    precision; features stay within fp16 round-off of the fp32 model and the forward pass runs end to end"""
    m = _model(fx, "cuda")
    im0, im1 = _images(fx, "cuda")
    with torch.no_grad():
        ref = m.features({"image0": im0, "image1": im1})
        from casmtr_amd.model import casmtr4c as mod
        mod._CONV_DTYPE[0] = torch.float16
        try:
            got = m.features({"image0": im0, "image1": im1})
        finally:
            mod._CONV_DTYPE[0] = None
    for (a0, a1), (b0, b1) in zip(ref, got):
        for a, b in ((a0, b0), (a1, b1)):
            assert b.dtype == torch.float32
            rel = float((a - b).abs().max() / a.abs().max())
            assert 0 < rel < 2e-2, rel
    m.conv_dtype = torch.float16
    out = m({"image0": im0, "image1": im1})
    assert out["mkpts0_f"].shape == out["mkpts1_f"].shape and out["mkpts0_f"].shape[1] == 2
    assert mod._CONV_DTYPE[0] is None
    with pytest.raises(ValueError):
        m({"image0": im0[..., :100], "image1": im1[..., :100]})


# ---------------------------------------------------------------------------------------------------------------------
# the indoor model (cascade_quadtree_stage3.py): ResNet-FPN + ladder, 8 QuadTree layers, POLA, relative position bias, no NMS
FIXI = os.path.join(HERE, "golden", "model_indoor_london_bridge.npz")


@pytest.fixture(scope="module")
def fxi():
    z = np.load(FIXI)
    return {k: z[k] for k in z.files}


def _model_indoor(fxi, device):
    from casmtr_amd.model import CasMTRIndoor4c, indoor_4c_config
    c = indoor_4c_config()
    thr = fxi["thresholds"]
    c["match_coarse"]["thr"] = float(thr[0])
    c["match_cascade"].update(test_thr=float(thr[1]), pre_thr=[float(thr[2])] * 2, double_check=bool(thr[3]))
    return _load_deterministic(CasMTRIndoor4c(c).eval()).to(device)


def test_state_dict_layout_indoor_matches_reference():
    from casmtr_amd.model import CasMTRIndoor4c
    with open(os.path.join(HERE, "golden", "model_indoor_state_keys.json")) as f:
        ref = json.load(f)
    mine = {k: list(v.shape) for k, v in CasMTRIndoor4c().state_dict().items()}
    assert sorted(mine) == sorted(ref)
    assert all(mine[k] == ref[k] for k in ref)


def _indoor_torch_parts(fxi, device, tol):
    """backbone, ladder and fine stage are torch ops only: they run on the CPU as well"""
    m = _model_indoor(fxi, device)
    im = [torch.from_numpy(fxi[k]).to(device).float() / 255.0 for k in ("image0", "image1")]
    data = {"image0": im[0], "image1": im[1]}
    with torch.no_grad():
        x, f8, f4, ff = m.features(data)
        _close(f8[:, ::4, ::2, ::2], fxi["bb_f8_sub"], tol, "1/8 features")
        _close(f4[:, ::4, ::2, ::2], fxi["bb_f4_sub"], tol, "1/4 features")
        _close(ff[:, ::4, ::4, ::4], fxi["bb_ff_sub"], tol, "1/2 features")
        l4, lf = m.ladder(x, f4, ff)
        _close(l4[:, ::4, ::2, ::2], fxi["lad_4_sub"], tol, "ladder 1/4")
        _close(lf[:, ::4, ::4, ::4], fxi["lad_f_sub"], tol, "ladder 1/2")
        t4 = torch.from_numpy(fxi["t4"]).to(device).float()
        t = lambda k, dt=torch.int64: torch.from_numpy(fxi[k].astype(np.int64) if dt == torch.int64 else fxi[k]).to(device).to(dt)
        data["stage_4c"] = {"b_ids": t("m4_b_ids"), "i_ids": t("m4_i_ids"), "j_ids": t("m4_j_ids"), "m_bids": t("m4_b_ids"),
                            "mconf": t("m4_mconf", torch.float32), "mkpts0_c": t("m4_mkpts0_c", torch.float32),
                            "mkpts1_c": t("m4_mkpts1_c", torch.float32)}
        m.fine_stage(lf[:1], lf[1:], t4[:1], t4[1:], data)
    assert len(fxi["mkpts1_f"]) >= 50
    _close(data["mkpts0_f"], fxi["mkpts0_f"], 0, "mkpts0_f")
    assert float((data["mkpts1_f"].cpu() - torch.from_numpy(fxi["mkpts1_f"])).abs().max()) < 1e-2   # pixels


def test_indoor_backbone_ladder_fine_cpu(fxi):
    _indoor_torch_parts(fxi, "cpu", 2e-4)


@pytest.mark.gpu
def test_indoor_backbone_ladder_fine_gpu(fxi):
    _indoor_torch_parts(fxi, "cuda", 1e-3)


@pytest.mark.gpu
def test_indoor_stages_on_reference_tensors(fxi):
    """8 QuadTree layers with top-k [32,16,16] on the fixture's fp16-exact 1/8 features; then ladder + UpBlock + POLA + cascade
    cross-attention WITH the learned relative position bias on the reference's 1/8 tokens and argmax; then the matcher (no NMS)
    on the reference's 1/4 tokens"""
    m = _model_indoor(fxi, "cuda")
    im = [torch.from_numpy(fxi[k]).cuda().float() / 255.0 for k in ("image0", "image1")]
    data = {"image0": im[0], "image1": im[1]}
    idx = lambda k: torch.from_numpy(fxi[k].astype(np.int64)).cuda()
    with torch.no_grad():
        x, f8, f4, ff = m.features(data)
        t0, t1 = m.coarse_stage(torch.from_numpy(fxi["f8"]).cuda().float(), data)
        _close(torch.cat([t0, t1]), fxi["t8"].astype(np.float32), 3e-3, "1/8 tokens", frac=0.99)
        t8 = torch.from_numpy(fxi["t8"]).cuda().float()
        data["stage_8c"] = {"next_idx_c01": idx("m8_next_idx_c01"), "next_idx_c10": idx("m8_next_idx_c10"),
                            "next_conf_c01": torch.from_numpy(fxi["m8_next_conf_c01"]).cuda(), "next_conf_c01_s": None}
        t4_0, t4_1, _, _ = m.cascade_stage(x, f4, ff, t8[:1], t8[1:], data)
    _close(torch.cat([t4_0, t4_1]), fxi["t4"].astype(np.float32), 3e-3, "1/4 tokens")
    from casmtr_amd import ops
    t4 = torch.from_numpy(fxi["t4"]).cuda().float()
    H4, W4 = data["hw0_4c"]
    wi = [ops.WindowIndex(ops.window_warp_idx(data["stage_8c"][k], H4 // 2, W4 // 2, 5), (H4, W4), (H4, W4), 1)
          for k in ("next_idx_c01", "next_idx_c10")]
    m.cascade_matching_4c(t4[:1].contiguous(), t4[1:].contiguous(), wi[0], wi[1], data, level="4c", pre_level="8c")
    s4 = data["stage_4c"]
    assert len(fxi["m4_i_ids"]) >= 50
    assert s4["i_ids"].cpu().tolist() == fxi["m4_i_ids"].astype(np.int64).tolist()
    assert s4["j_ids"].cpu().tolist() == fxi["m4_j_ids"].astype(np.int64).tolist()
    _close(s4["mconf"], fxi["m4_mconf"], 1e-4, "mconf")


@pytest.mark.gpu
def test_whole_forward_indoor(fxi):
    m = _model_indoor(fxi, "cuda")
    im = [torch.from_numpy(fxi[k]).cuda().float() / 255.0 for k in ("image0", "image1")]
    data = m({"image0": im[0], "image1": im[1]})
    mine = {(int(a[0]), int(a[1])): b for a, b in zip(data["mkpts0_f"].cpu().tolist(), data["mkpts1_f"].cpu())}
    ref0, ref1 = fxi["mkpts0_f"], torch.from_numpy(fxi["mkpts1_f"])
    hit = sum(1 for a, b in zip(ref0.tolist(), ref1) if (int(a[0]), int(a[1])) in mine
              and float((mine[(int(a[0]), int(a[1]))] - b).abs().max()) < 0.25)
    assert hit >= 0.85 * len(ref0), f"{hit} of {len(ref0)} reference matches reproduced ({len(mine)} found)"


@pytest.mark.gpu
@pytest.mark.parametrize("cls,size,B", [("CasMTR4c", (480, 640), 3), ("CasMTR4c", (256, 256), 1), ("CasMTR2c", (320, 256), 2),
                                        ("CasMTRIndoor4c", (480, 640), 2), ("CasMTRIndoor4c", (256, 384), 1)])
def test_models_run_at_other_sizes(cls, size, B):
    """grids that are not multiples of the 7 x 7 attention windows, odd coarsest levels, several batch sizes: the forward pass runs,
    is finite and puts every match inside the images; two runs agree up to the run-to-run round-off of the library GEMMs /
    convolutions in the glue (the hot path itself is bit-reproducible: test_full_size_run_to_run_determinism)"""
    import casmtr_amd.model as M
    torch.manual_seed(0)
    m = getattr(M, cls)()
    for k, c in (("match_coarse", m.config["match_coarse"]), ("match_cascade", m.config["match_cascade"])):
        c.update(thr=0.0) if k == "match_coarse" else c.update(test_thr=0.0, pre_thr=[0.0, 0.0], double_check=False)
    if "match_cascade_2c" in m.config:
        m.config["match_cascade_2c"].update(test_thr=0.0, pre_thr=[0.0, 0.0], double_check=False)
    m = getattr(M, cls)(m.config).eval().cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    im0, im1 = (torch.rand((B, 3, *size), device="cuda", generator=g) for _ in range(2))
    a = m({"image0": im0, "image1": im1})
    b = m({"image0": im0, "image1": im1})
    assert a["mkpts0_f"].shape[0] > 0
    for k in ("mkpts0_f", "mkpts1_f", "expec_f"):
        assert torch.isfinite(a[k]).all()
    assert abs(a["mkpts0_f"].shape[0] - b["mkpts0_f"].shape[0]) <= 0.01 * a["mkpts0_f"].shape[0] + 2
    if a["mkpts0_f"].shape == b["mkpts0_f"].shape and torch.equal(a["mkpts0_f"], b["mkpts0_f"]):
        assert float((a["mkpts1_f"] - b["mkpts1_f"]).abs().max()) < 0.05   # pixels
    assert int(a["m_bids"].max()) < B
    H, W = size
    assert (a["mkpts0_f"][:, 0] >= 0).all() and (a["mkpts0_f"][:, 0] < W).all() and (a["mkpts0_f"][:, 1] < H).all()
    assert (a["mkpts1_f"] > -8).all() and (a["mkpts1_f"][:, 0] < W + 8).all() and (a["mkpts1_f"][:, 1] < H + 8).all()


@pytest.mark.gpu
def test_model_respects_padding_masks():
    """mask{0,1}_origin (MegaDepth-style zero padding at the bottom / right): no match may start or end in a padded region"""
    from casmtr_amd.model import CasMTR4c, outdoor_4c_config
    torch.manual_seed(0)
    c = outdoor_4c_config()
    c["match_coarse"]["thr"] = 0.0
    c["match_cascade"].update(test_thr=0.0, pre_thr=[0.0], double_check=False)
    m = CasMTR4c(c).eval().cuda()
    g = torch.Generator(device="cuda").manual_seed(2)
    B, H, W = 2, 320, 384
    im0, im1 = (torch.rand((B, 3, H, W), device="cuda", generator=g) for _ in range(2))
    m0 = torch.ones((B, H, W), dtype=torch.bool, device="cuda")
    m1 = torch.ones((B, H, W), dtype=torch.bool, device="cuda")
    m0[:, :, 288:] = False      # image 0: right quarter is padding
    m1[:, 224:, :] = False      # image 1: bottom 30 % is padding
    im0 = im0 * m0[:, None]
    im1 = im1 * m1[:, None]
    out = m({"image0": im0, "image1": im1, "mask0_origin": m0, "mask1_origin": m1})
    assert out["mkpts0_f"].shape[0] > 100
    assert float(out["mkpts0_f"][:, 0].max()) < 288
    assert float(out["stage_4c"]["mkpts1_c"][:, 1].max()) < 224
    assert float(out["mkpts1_f"][:, 1].max()) < 224 + 8


# ---- GroupAttention.forward_mask's padding quirk: fixture from the reference module, tests/golden/gen_golden_window_attn.py
def _window_attn_case(H, W, device):
    import golden_inputs as gi
    from casmtr_amd.model.casmtr4c import _WindowAttention
    m = _WindowAttention(gi.WINDOW_ATTN_DIM, gi.WINDOW_ATTN_HEADS, gi.WINDOW_ATTN_WS, qkv_bias=True).to(device).eval()
    m.load_state_dict(gi.window_attn_weights())
    return m, gi.window_attn_tokens(H, W).to(device), gi.WINDOW_ATTN_GRIDS


@pytest.mark.parametrize("hw", [(14, 20), (20, 14), (10, 17), (14, 21)], ids=lambda t: f"{t[0]}x{t[1]}")
def test_window_attention_padding_quirk_cpu(hw):
    """exactly one grid side a multiple of ws: the reference's mask is all ones (no masking, padded keys take part); both or neither:
    the usual -1000 mask -- torch path of the harness against the reference module's outputs"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "window_attn_padding.npz"))
    m, x, _ = _window_attn_case(*hw, "cpu")
    with torch.no_grad():
        y = m(x, *hw)
    assert float((y - torch.from_numpy(g[f"out_{hw[0]}x{hw[1]}"])).abs().max()) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(14, 20), (20, 14), (10, 17), (14, 21)], ids=lambda t: f"{t[0]}x{t[1]}")
def test_window_attention_padding_quirk_gpu(hw):
    """the same on the GPU: the quirk sizes take the unmasked padded path, the others the HIP window kernel"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "window_attn_padding.npz"))
    m, x, _ = _window_attn_case(*hw, "cuda:0")
    with torch.no_grad():
        y = m(x, *hw)
    assert float((y.cpu() - torch.from_numpy(g[f"out_{hw[0]}x{hw[1]}"])).abs().max()) < 1e-4
