"""Seeded synthetic inputs shared by tests/golden/gen_golden.py (which feeds them to the reference python) and by
the parity tests (which feed the same arrays to the oracle and to the HIP path).  numpy only."""
import zlib

import numpy as np

CASES = {
    # a1 / a2 / a8 primitives
    "ops": {
        "k64_h8": dict(seed=11, B=1, N1=16, N2=256, K=64, H=8, D=32, WK=100, WC=128),
        "k100_h4": dict(seed=12, B=2, N1=16, N2=400, K=100, H=4, D=32, WK=100, WC=128),
        "k128_h2": dict(seed=13, B=1, N1=16, N2=144, K=128, H=2, D=32, WK=36, WC=64),
    },
    # QTAttB: finest grid, topks (coarsest first), heads
    "qtattb": {
        "g16x16": dict(seed=21, B=2, full=True, hw=(16, 16), nhead=8, D=32, topks=[8, 4, 2]),
        "g32x24_cross": dict(seed=22, B=1, hw=(32, 24), hw_k=(24, 32), nhead=8, D=32, topks=[8, 4, 2]),
        "g40x32": dict(seed=23, B=1, hw=(40, 32), nhead=8, D=32, topks=[32, 16, 8]),
        "g40x32_indoor": dict(seed=24, B=1, hw=(40, 32), nhead=8, D=32, topks=[32, 16, 16]),
    },
    # a13: variants no shipped config selects (QTAttA; QTAttGuided starts from externally supplied positions)
    "qtatt_variants": {
        "a_g16x16": dict(seed=61, B=1, hw=(16, 16), nhead=8, D=32, topks=[8, 4, 2], kind="A"),
        "guided_g16x16": dict(seed=62, B=1, hw=(16, 16), nhead=4, D=32, topks=[4, 4], kind="Guided"),
    },
    "cascade_attn": {
        "c16_f32": dict(seed=31, B=2, coarse_hw=(16, 16), nhead=4, D=32, ws=5),
        "c12x20_relpos": dict(seed=32, B=1, coarse_hw=(12, 20), nhead=4, D=32, ws=5, rel_pos=True),
        "c8_h2": dict(seed=33, B=1, coarse_hw=(8, 8), nhead=2, D=32, ws=5),
    },
    # §8 f.1: the callers QuadtreeAttention / CascadeQuadtreeAttention on [B,N,C] tokens.  `exact`: activations and
    # projection weights on a dyadic grid coarse enough that every partial sum of the q/k/v projections is exactly
    # representable -- the reference's BLAS and the k-ordered chain then agree bit for bit whatever their summation order.
    "quadtree_block": {
        "qa_g16_exact": dict(seed=71, kind="qta", B=2, hw=(16, 16), nhead=4, D=32, topks=[8, 4, 2], exact=True),
        "qa_g32x24_cross_bias": dict(seed=72, kind="qta", B=1, hw=(32, 24), hw1=(24, 32), nhead=4, D=32, topks=[8, 4, 2],
                                     exact=True, qkv_bias=True),
        "qa_g16_randn": dict(seed=73, kind="qta", B=1, hw=(16, 16), nhead=8, D=32, topks=[8, 4, 2], exact=False),
        "cqa_c8_exact": dict(seed=74, kind="cqa", B=2, coarse_hw=(8, 8), nhead=4, D=32, ws=5, exact=True),
    },
    "coarse_matching": {
        "g24": dict(seed=41, B=2, hw0=(24, 24), hw1=(24, 24), C=256),
        "g16_conf": dict(seed=44, B=1, hw0=(16, 16), hw1=(16, 16), C=256, store_conf=True),
        "g24x20_masks": dict(seed=42, B=2, hw0=(24, 20), hw1=(20, 24), C=256, masks=True, border_rm=2),
        "g16_border": dict(seed=43, B=1, hw0=(16, 16), hw1=(16, 16), C=256, border_rm=1, thr=0.05),
    },
    "cascade_matching": {
        "nms": dict(seed=51, B=2, coarse_hw=(12, 12), C=128, nms=True),
        "nonms_masks": dict(seed=52, B=2, coarse_hw=(12, 16), C=128, nms=False, masks=True, test_thr=0.1),
        "nms_nodc": dict(seed=53, B=1, coarse_hw=(10, 10), C=128, nms=True, double_check=False, border_rm=0),
        # §8 f.4: PostProcess 'local_window_nms' (top-k per non-overlapping window), unused by the shipped configs
        "local_window": dict(seed=54, B=2, coarse_hw=(12, 16), C=128, post=dict(method="local_window_nms", window_size=4, topk=2),
                             test_thr=0.05),
        # §8 f.4: PostProcess 'd2d' (as many positions as the max-pool NMS keeps, from the top of the detector score S_d2d)
        "d2d": dict(seed=55, B=2, coarse_hw=(12, 16), C=128, post=dict(method="d2d", window_size=5), test_thr=0.05, border_rm=0),
    },
}


def _rng(seed):
    return np.random.default_rng(seed)


def _randn(r, *shape):
    return r.standard_normal(shape, dtype=np.float32)


def _pyramid(x, levels=3):
    """avg_pool2d(k=2,s=2) pyramid, finest first (src/model/modules/quadtree_attention.py:81-89)."""
    out = [x]
    for _ in range(levels - 1):
        B, C, h, w = x.shape
        x = x.reshape(B, C, h // 2, 2, w // 2, 2).mean(axis=(3, 5), dtype=np.float32)
        out.append(np.ascontiguousarray(x))
    return out


def _valid_mask(B, h, w, r, lo=0.7):
    m = np.zeros((B, h, w), np.uint8)
    for b in range(B):
        vh = int(r.integers(int(h * lo), h + 1))
        vw = int(r.integers(int(w * lo), w + 1))
        m[b, :vh, :vw] = 1
    return m


def _warped_features(r, B, hw0, hw1, C, noise=0.35):
    """feat1 = feat0 moved by a smooth random permutation + noise, so a good share of matches are confident."""
    (h0, w0), (h1, w1) = hw0, hw1
    f0 = _randn(r, B, h0 * w0, C)
    f1 = _randn(r, B, h1 * w1, C)
    for b in range(B):
        n = min(h0 * w0, h1 * w1)
        src = r.permutation(h0 * w0)[: int(n * 0.6)]
        dst = r.permutation(h1 * w1)[: int(n * 0.6)]
        f1[b, dst] = f0[b, src] + noise * _randn(r, len(src), C)
    return f0, f1


def make_inputs(group, name):
    cfg = CASES[group][name]
    r = _rng(cfg["seed"])
    if group == "ops":
        B, N1, N2, K, H, D = (cfg[k] for k in ("B", "N1", "N2", "K", "H", "D"))
        return dict(
            q=_randn(r, B, N1, 4, H, D), key=_randn(r, B, N2, H, D), value=_randn(r, B, N2, H, D),
            idx=r.integers(0, N2, (B, N1, K, H), dtype=np.int64),
            wq=_randn(r, B, N1 * 4, cfg["WC"]), wkey=_randn(r, B, N2, cfg["WC"]),
            widx=r.integers(0, N2, (B, N1 * 4, cfg["WK"]), dtype=np.int64),
        )
    if group == "qtattb":
        B, (h, w), H, D = cfg["B"], cfg["hw"], cfg["nhead"], cfg["D"]
        hk, wk = cfg.get("hw_k", cfg["hw"])
        C = H * D
        q = _pyramid(_randn(r, B, C, h, w))
        k = _pyramid(_randn(r, B, C, hk, wk))
        v = _pyramid(_randn(r, B, C, hk, wk))
        return dict(queries=q, keys=k, values=v, weight=_randn(r, 3))
    if group == "qtatt_variants":
        B, (h, w), H, D = cfg["B"], cfg["hw"], cfg["nhead"], cfg["D"]
        C, n = H * D, len(cfg["topks"])
        q, k, v = (_pyramid(_randn(r, B, C, h, w), n) for _ in range(3))
        out = dict(queries=q, keys=k, values=v, weight=_randn(r, n))
        if cfg["kind"] == "Guided":   # (row, col) on the half-resolution grid of the coarsest level
            hc, wc = h >> (n - 1), w >> (n - 1)
            pos = np.stack([r.integers(0, hc // 2, (B, (hc // 2) * (wc // 2), cfg["topks"][0], H)),
                            r.integers(0, wc // 2, (B, (hc // 2) * (wc // 2), cfg["topks"][0], H))]).astype(np.int64)
            out["topk_pos"] = pos
        return out
    if group == "cascade_attn":
        B, (hc, wc), H, D = cfg["B"], cfg["coarse_hw"], cfg["nhead"], cfg["D"]
        C, h, w = H * D, hc * 2, wc * 2
        out = dict(q=_randn(r, B, C, h, w), k=_randn(r, B, C, h, w), v=_randn(r, B, C, h, w),
                   coarse_idx=r.integers(0, hc * wc, (B, hc * wc), dtype=np.int64))
        if cfg.get("rel_pos"):
            out["rel_pos"] = _randn(r, B, H, h * w, 4 * cfg["ws"] ** 2)
        return out
    if group == "quadtree_block":
        H, D, B = cfg["nhead"], cfg["D"], cfg["B"]
        C = H * D
        if cfg["kind"] == "qta":
            (h, w), (h1, w1) = cfg["hw"], cfg.get("hw1", cfg["hw"])
        else:
            h = h1 = cfg["coarse_hw"][0] * 2
            w = w1 = cfg["coarse_hw"][1] * 2
        dy = (lambda a, s, q: np.round(a * s) / q) if cfg["exact"] else (lambda a, s, q: a * (s / q))
        out = dict(x=dy(_randn(r, B, h * w, C), 4, 8).astype(np.float32),
                   target=dy(_randn(r, B, h1 * w1, C), 4, 8).astype(np.float32),
                   wq=dy(_randn(r, C, C), 4, 16).astype(np.float32), wk=dy(_randn(r, C, C), 4, 16).astype(np.float32),
                   wv=dy(_randn(r, C, C), 4, 16).astype(np.float32), wp=(_randn(r, C, C) / np.sqrt(C)).astype(np.float32),
                   bp=_randn(r, C) * np.float32(0.1))
        if cfg.get("qkv_bias"):
            for n in ("bq", "bk", "bv"):
                out[n] = dy(_randn(r, C), 4, 8).astype(np.float32)
        if cfg["kind"] == "qta":
            out["weight"] = _randn(r, 3)
        else:
            hc, wc = cfg["coarse_hw"]
            out["coarse_idx"] = r.integers(0, hc * wc, (B, hc * wc), dtype=np.int64)
        return out
    if group == "coarse_matching":
        f0, f1 = _warped_features(r, cfg["B"], cfg["hw0"], cfg["hw1"], cfg["C"])
        out = dict(feat0=f0, feat1=f1)
        if cfg.get("masks"):
            out["mask0"] = _valid_mask(cfg["B"], *cfg["hw0"], r)
            out["mask1"] = _valid_mask(cfg["B"], *cfg["hw1"], r)
        return out
    if group == "cascade_matching":
        B, (hc, wc), C = cfg["B"], cfg["coarse_hw"], cfg["C"]
        h, w = hc * 2, wc * 2
        # image1 = image0 shifted by a couple of pixels + noise -> windows around the coarse match contain the match
        f0 = _randn(r, B, h, w, C)
        dy, dx = 1, 2
        f1 = _randn(r, B, h, w, C)
        f1[:, dy:, dx:] = f0[:, : h - dy, : w - dx] + 0.5 * _randn(r, B, h - dy, w - dx, C)
        f0 *= 2.0
        f1 *= 2.0
        ys, xs = np.meshgrid(np.arange(hc), np.arange(wc), indexing="ij")
        c01 = (np.clip(ys + 0, 0, hc - 1) * wc + np.clip(xs + 1, 0, wc - 1)).reshape(1, -1).repeat(B, 0)
        c10 = (np.clip(ys - 0, 0, hc - 1) * wc + np.clip(xs - 1, 0, wc - 1)).reshape(1, -1).repeat(B, 0)
        # sprinkle some random coarse matches so borders / far windows are exercised too
        rnd = r.random((B, hc * wc)) < 0.15
        c01 = np.where(rnd, r.integers(0, hc * wc, (B, hc * wc)), c01).astype(np.int64)
        out = dict(feat0=f0.reshape(B, h * w, C), feat1=f1.reshape(B, h * w, C), coarse_idx01=c01,
                   coarse_idx10=c10.astype(np.int64), pre_conf=r.random((B, hc * wc), dtype=np.float32))
        if cfg.get("masks"):
            out["mask0"] = _valid_mask(B, h, w, r, lo=0.75)
            out["mask1"] = _valid_mask(B, h, w, r, lo=0.75)
        return out
    raise KeyError(group)


def make_grad_inputs(group, name):
    """Seeded upstream gradients (and the extra forward inputs the backward fixtures need), a12 / f2.
    Separate generator (seed + 1000) so that the forward fixtures' inputs and checksums stay what they were."""
    cfg = CASES[group][name]
    r = _rng(cfg["seed"] + 1000)
    if group == "ops":
        B, N1, N2, K, H, D = (cfg[k] for k in ("B", "N1", "N2", "K", "H", "D"))
        return dict(g_score=_randn(r, B, N1, 4, K, H), agg_score=r.random((B, N1, 4, K, H), dtype=np.float32),
                    g_msg=_randn(r, B, N1, 4, H, D), g_window=_randn(r, B, N1 * 4, cfg["WK"]))
    if group == "qtattb":
        B, (h, w), H, D = cfg["B"], cfg["hw"], cfg["nhead"], cfg["D"]
        return dict(g_final=_randn(r, B, h * w, H, D))
    if group == "cascade_attn":
        B, (hc, wc), H, D = cfg["B"], cfg["coarse_hw"], cfg["nhead"], cfg["D"]
        return dict(g_message=_randn(r, B, 4 * hc * wc, H * D))
    raise KeyError(group)


GRAD_CASES = {"ops": list(CASES["ops"]), "qtattb": ["g16x16", "g32x24_cross"], "cascade_attn": ["c16_f32", "c12x20_relpos"]}


def checksum(inp):
    """crc32 over every input array, stored in the fixture to detect RNG drift."""
    c = 0
    for k in sorted(inp):
        vs = inp[k] if isinstance(inp[k], (list, tuple)) else [inp[k]]
        for v in vs:
            c = zlib.crc32(np.ascontiguousarray(v).tobytes(), c)
    return np.array([c], dtype=np.int64)


# ---------------------------------------------------------------------------------------------------------------------
# f.3 model harness: one deterministic state dict for the whole CasMTR-4c, rebuilt from the parameter names alone so that
# the fixture generator (reference model, CPU) and the GPU test (casmtr_amd.model) hold identical weights without storing
# 30 M floats.  Scales are chosen so that activations stay O(1) through the random network.
def model_state(shapes):
    """shapes: {state-dict key: shape} -> {key: float32 / int64 ndarray}.  Integer entries (num_batches_tracked, window) are
    skipped: both sides keep their own, identical, constructor values."""
    out = {}
    for key, shape in shapes.items():
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked" or key.endswith(".window"):
            continue
        r = np.random.RandomState(zlib.crc32(key.encode()) & 0x7FFFFFFF)
        shape = tuple(shape)
        n = r.standard_normal(shape).astype(np.float32)
        if leaf == "running_var":
            v = 1.0 + 0.25 * np.abs(n)
        elif leaf == "running_mean":
            v = 0.05 * n
        elif leaf == "bias":
            v = 0.05 * n
        elif len(shape) == 1:                       # LayerNorm / BatchNorm gains, QTAttB level weights
            v = 1.0 + 0.1 * n
        else:
            fan_in = int(np.prod(shape[1:]))
            v = n * (1.0 / np.sqrt(fan_in))
        out[key] = np.ascontiguousarray(v, dtype=np.float32)
    return out


# ---- GroupAttention.forward_mask padding quirk (tests/golden/gen_golden_window_attn.py, tests/test_model_harness.py)
WINDOW_ATTN_GRIDS = [(14, 20), (20, 14), (10, 17), (14, 21)]   # ws = 7: pad (bottom, right) = (0, 1), (1, 0), (4, 4), (0, 0)
WINDOW_ATTN_DIM, WINDOW_ATTN_HEADS, WINDOW_ATTN_WS = 64, 2, 7


def window_attn_weights(seed=3):
    import torch
    g = torch.Generator().manual_seed(seed)
    d = WINDOW_ATTN_DIM
    return {"qkv.weight": torch.randn(3 * d, d, generator=g) / d ** 0.5, "qkv.bias": torch.randn(3 * d, generator=g),
            "proj.weight": torch.randn(d, d, generator=g) / d ** 0.5, "proj.bias": torch.randn(d, generator=g)}


def window_attn_tokens(H, W, seed=4):
    import torch
    g = torch.Generator().manual_seed(seed + 100 * H + W)
    return torch.randn(2, H * W, WINDOW_ATTN_DIM, generator=g)
