"""CPU oracle for the CasMTR cascaded-matching hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; nothing under ``casmtr_amd/`` does (the product path fails loudly without its HIP library).

The arithmetic lives in ``casmtr_oracle.c`` (plain C, each function cites the reference file:line it
restates).  This module is the numpy/ctypes front end plus the two pieces of pure index glue that the
reference keeps in python (``QTAttB.forward``'s level loop, ``mkpts`` scaling).

Parity status: pinned against ``tests/golden/*.npz`` -- outputs of the reference python itself, generated in
the build container by ``tests/golden/gen_golden.py`` (the reference ships no tests of its own).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcasmtr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile casmtr_oracle.c with gcc (seconds).  Idempotent."""
    src = os.path.join(_HERE, "casmtr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libcasmtr_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_dual_softmax.restype = C.c_int64
        _lib.orc_nms_select.restype = C.c_int64
    return _lib


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int64)


def _u8(a):
    return None if a is None else np.ascontiguousarray(a).astype(np.uint8)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _ci(*xs):
    return [C.c_int(int(x)) for x in xs]


# ---------------------------------------------------------------------------------------------------- ops
def qta_score_fwd(q, key, idx):
    q, key, idx = _f(q), _f(key), _i(idx)
    B, N1, _, H, D = q.shape
    N2, K = key.shape[1], idx.shape[2]
    out = np.empty((B, N1, 4, K, H), np.float32)
    lib().orc_qta_score_fwd(_p(q), _p(key), _p(idx), _p(out), *_ci(B, N1, N2, K, H, D))
    return out


def qta_score_bwd(grad, q, key, idx):
    grad, q, key, idx = _f(grad), _f(q), _f(key), _i(idx)
    B, N1, _, H, D = q.shape
    N2, K = key.shape[1], idx.shape[2]
    dq, dk = np.empty_like(q), np.empty_like(key)
    lib().orc_qta_score_bwd(_p(grad), _p(q), _p(key), _p(idx), _p(dq), _p(dk), *_ci(B, N1, N2, K, H, D))
    return dq, dk


def qta_value_agg_fwd(score, value, idx):
    score, value, idx = _f(score), _f(value), _i(idx)
    B, N, K, H = score.shape
    M, D = value.shape[1], value.shape[3]
    out = np.empty((B, N, H, D), np.float32)
    lib().orc_qta_value_agg_fwd(_p(score), _p(value), _p(idx), _p(out), *_ci(B, N, K, H, M, D))
    return out


def qta_value_agg_bwd(grad_out, score, value, idx):
    grad_out, score, value, idx = _f(grad_out), _f(score), _f(value), _i(idx)
    B, N, K, H = score.shape
    M, D = value.shape[1], value.shape[3]
    gs, gv = np.empty_like(score), np.empty_like(value)
    lib().orc_qta_value_agg_bwd(_p(grad_out), _p(score), _p(value), _p(idx), _p(gs), _p(gv), *_ci(B, N, K, H, M, D))
    return gs, gv


def window_score_fwd(q, key, idx):
    q, key, idx = _f(q), _f(key), _i(idx)
    B, N1, Cc = q.shape
    N2, K = key.shape[1], idx.shape[2]
    out = np.empty((B, N1, K), np.float32)
    lib().orc_window_score_fwd(_p(q), _p(key), _p(idx), _p(out), *_ci(B, N1, N2, K, Cc))
    return out


def window_score_bwd(grad, q, key, idx):
    grad, q, key, idx = _f(grad), _f(q), _f(key), _i(idx)
    B, N1, Cc = q.shape
    N2, K = key.shape[1], idx.shape[2]
    dq, dk = np.empty_like(q), np.empty_like(key)
    lib().orc_window_score_bwd(_p(grad), _p(q), _p(key), _p(idx), _p(dq), _p(dk), *_ci(B, N1, N2, K, Cc))
    return dq, dk


def qta_coarse_level(q, k, v, topk, want_A=False):
    """q [B,L,H,D], k/v [B,S,H,D] -> message, topk_score, topk_idx (, A)."""
    q, k, v = _f(q), _f(k), _f(v)
    B, L, H, D = q.shape
    S = k.shape[1]
    temp = np.float32(1.0 / D ** 0.5)
    msg = np.empty((B, L, H, D), np.float32)
    ts = np.empty((B, L, topk, H), np.float32)
    ti = np.empty((B, L, topk, H), np.int64)
    A = np.empty((B, L, S, H), np.float32) if want_A else None
    lib().orc_qta_coarse_level(_p(q), _p(k), _p(v), C.c_float(temp), C.c_int(topk), _p(msg), _p(ts), _p(ti), _p(A),
                               *_ci(B, L, S, H, D))
    return (msg, ts, ti, A) if want_A else (msg, ts, ti)


def qta_fine_level(q, key, value, prev_idx, hw0, hw1, topk, w_level=1.0, acc_in=None, want_A=False):
    """q [B,h0*w0,H,D] raster, key/value [B,h1*w1,H,D], prev_idx [B,L/4,Kp,H] -> dict of raster-order outputs."""
    q, key, value, prev_idx, acc_in = _f(q), _f(key), _f(value), _i(prev_idx), _f(acc_in)
    B, L, H, D = q.shape
    (h0, w0), (h1, w1) = hw0, hw1
    Kp = prev_idx.shape[2]
    temp = np.float32(1.0 / D ** 0.5)
    msg = np.empty((B, L, H, D), np.float32)
    acc = np.empty((B, L, H, D), np.float32)
    ts = np.empty((B, L, max(topk, 1), H), np.float32)
    ti = np.empty((B, L, max(topk, 1), H), np.int64)
    A = np.empty((B, L, 4 * Kp, H), np.float32) if want_A else None
    lib().orc_qta_fine_level(_p(q), _p(key), _p(value), _p(prev_idx), C.c_float(temp), C.c_int(topk),
                             C.c_float(w_level), _p(acc_in), _p(msg), _p(acc), _p(ts), _p(ti), _p(A),
                             *_ci(B, h0, w0, h1, w1, H, D, Kp))
    return dict(message=msg, acc=acc, topk_score=ts, topk_idx=ti, A=A)


def to_tokens(x_nchw, nhead):
    """[B,C,h,w] -> [B,h*w,H,D]  (modules/quadtree_attention.py:165-167)."""
    B, Cc, h, w = x_nchw.shape
    return np.ascontiguousarray(np.transpose(x_nchw, (0, 2, 3, 1)).reshape(B, h * w, nhead, Cc // nhead))


def qtattb_forward(queries, keys, values, weight, nhead, topks):
    """QTAttB.forward restated over the C kernels (modules/quadtree_attention.py:231-286).

    queries/keys/values: lists of [B,C,h,w] numpy arrays, finest first.  Returns (final [B,L,H,D], per-level dicts).
    """
    w = np.asarray(weight, np.float32)
    e = np.exp(w - w.max())
    wsm = (e / e.sum()).astype(np.float32)  # torch.softmax(self.weight, dim=0), :264
    levels = []
    acc = None
    prev_idx = None
    for i, (qq, kk, vv) in enumerate(zip(reversed(queries), reversed(keys), reversed(values))):
        h0, w0 = qq.shape[2:]
        h1, w1 = kk.shape[2:]
        qt, kt, vt = to_tokens(qq, nhead), to_tokens(kk, nhead), to_tokens(vv, nhead)
        if i == 0:
            msg, ts, ti = qta_coarse_level(qt, kt, vt, topks[0])
            acc = msg * wsm[0]
            levels.append(dict(message=msg, topk_score=ts, topk_idx=ti, acc=acc))
        else:
            out = qta_fine_level(qt, kt, vt, prev_idx, (h0, w0), (h1, w1), topks[i], wsm[i], acc)
            acc = out["acc"]
            levels.append(out)
        prev_idx = levels[-1]["topk_idx"]
    return acc, levels


def cascade_attn(q, key, value, topk_pos, hw0, hw1, nhead, dilated=1, rel_pos=None, want_A=False):
    """q [B,h0*w0,C], key/value [B,h1*w1,C], topk_pos [B,L/4,KW,2] -> message [B,L,C], up_idx [B,L,4KW]."""
    q, key, value, topk_pos, rel_pos = _f(q), _f(key), _f(value), _i(topk_pos), _f(rel_pos)
    B, L, Cc = q.shape
    (h0, w0), (h1, w1) = hw0, hw1
    KW = topk_pos.shape[2]
    D = Cc // nhead
    temp = np.float32(1.0 / D ** 0.5)
    msg = np.empty((B, L, Cc), np.float32)
    up = np.empty((B, L, 4 * KW), np.int64)
    A = np.empty((B, L, 4 * KW, nhead), np.float32) if want_A else None
    lib().orc_cascade_attn(_p(q), _p(key), _p(value), _p(topk_pos), _p(rel_pos), C.c_float(temp), C.c_int(dilated),
                           _p(msg), _p(up), _p(A), *_ci(B, h0, w0, h1, w1, nhead, D, KW))
    return (msg, up, A) if want_A else (msg, up)


def dual_softmax(feat0, feat1, hw0, hw1, temperature=0.1, thr=0.2, border_rm=0, mask0=None, mask1=None,
                 valid_hw=None, recip=False, want_conf=False):
    feat0, feat1, mask0, mask1 = _f(feat0), _f(feat1), _u8(mask0), _u8(mask1)
    B, L, Cc = feat0.shape
    S = feat1.shape[1]
    vh = None if valid_hw is None else np.ascontiguousarray(valid_hw, dtype=np.int32)
    conf = np.empty((B, L, S), np.float32) if want_conf else None
    ni01, nc01 = np.empty((B, L), np.int64), np.empty((B, L), np.float32)
    ni10, nc10 = np.empty((B, S), np.int64), np.empty((B, S), np.float32)
    bi, ii, ji = (np.empty(B * L, np.int64) for _ in range(3))
    mc = np.empty(B * L, np.float32)
    n = lib().orc_dual_softmax(_p(feat0), _p(feat1), _p(mask0), _p(mask1), C.c_float(temperature), C.c_int(int(recip)),
                               C.c_float(thr), C.c_int(border_rm), _p(vh), *_ci(hw0[0], hw0[1], hw1[0], hw1[1]),
                               _p(conf), _p(ni01), _p(nc01), _p(ni10), _p(nc10), _p(bi), _p(ii), _p(ji), _p(mc),
                               *_ci(B, L, S, Cc))
    return dict(conf_matrix=conf, next_idx_c01=ni01, next_conf_c01=nc01, next_idx_c10=ni10, next_conf_c10=nc10,
                b_ids=bi[:n].copy(), i_ids=ii[:n].copy(), j_ids=ji[:n].copy(), mconf=mc[:n].copy())


def window_match(feat_q, feat_k, idx, temperature=1.0, mask_q=None, mask_k=None, recip=False, want_conf=True):
    feat_q, feat_k, idx, mask_q, mask_k = _f(feat_q), _f(feat_k), _i(idx), _u8(mask_q), _u8(mask_k)
    B, N, Cc = feat_q.shape
    M, K = feat_k.shape[1], idx.shape[2]
    conf = np.empty((B, N, K), np.float32) if want_conf else None
    nc, ni = np.empty((B, N), np.float32), np.empty((B, N), np.int64)
    lib().orc_window_match(_p(feat_q), _p(feat_k), _p(idx), _p(mask_q), _p(mask_k), C.c_float(temperature),
                           C.c_int(int(recip)), _p(conf), _p(nc), _p(ni), *_ci(B, N, M, K, Cc))
    return dict(conf_matrix=conf, next_conf=nc, next_idx=ni)


def local_window_topk_mask(conf, hw, window_size, topk):
    """PostProcess 'local_window_nms' (post_processing.py:76-93): the top-`topk` confidences of every non-overlapping
    window_size x window_size tile survive.  (value desc, position asc); conf [B,h*w] -> bool [B,h*w]."""
    conf = _f(conf)
    B = conf.shape[0]
    h, w = hw
    ws = window_size
    t = conf.reshape(B, h // ws, ws, w // ws, ws).transpose(0, 1, 3, 2, 4).reshape(B, -1, ws * ws)
    order = np.argsort(-t, axis=2, kind="stable")[:, :, :topk]
    keep = np.zeros_like(t, dtype=bool)
    np.put_along_axis(keep, order, True, axis=2)
    return np.ascontiguousarray(keep.reshape(B, h // ws, w // ws, ws, ws).transpose(0, 1, 3, 2, 4).reshape(B, h * w))


def d2d_scores(feat, hw):
    """CascadeMatching.forward, 'd2d' branch (cascade_matching.py:88-104): S_d2d [B,(h/4)*(w/4),1] from feat_c0 [B,h*w,C].
    std over channels (unbiased, torch.std) of feat / sqrt(C) at every 4th position (F.interpolate nearest, x0.25) times the channel
    norm of the depth-wise 5x5 response (kernel -1/25, centre 24, stride 4, zero padding 2), min-max normalised over the whole batch."""
    feat = _f(feat)
    B, N, Cc = feat.shape
    h, w = hw
    x = (feat / np.float32(Cc ** .5)).reshape(B, h, w, Cc).astype(np.float32)
    s_as = x[:, ::4, ::4].std(axis=-1, ddof=1, dtype=np.float64).astype(np.float32)
    xp = np.pad(x, ((0, 0), (2, 2), (2, 2), (0, 0)))
    ho, wo = (h + 4 - 5) // 4 + 1, (w + 4 - 5) // 4 + 1
    resp = np.zeros((B, ho, wo, Cc), np.float64)
    for dy in range(5):
        for dx in range(5):
            kv = 24.0 if (dy, dx) == (2, 2) else -1.0 / 25.0
            resp += kv * xp[:, dy:dy + 4 * ho:4, dx:dx + 4 * wo:4].astype(np.float64)
    s_rs = np.sqrt((resp.astype(np.float32).astype(np.float64) ** 2).sum(-1)).astype(np.float32)
    s_rs = (s_rs - s_rs.min()) / (s_rs.max() - s_rs.min())
    return (s_as.reshape(B, -1, 1) * s_rs.reshape(B, -1, 1)).astype(np.float32)


def d2d_mask(next_conf, s_d2d, hw, window_size):
    """PostProcess 'd2d' (post_processing.py:122-143) before the threshold: per pair as many positions as the max-pool NMS keeps, taken
    from the top of S_d2d (sub-grid position (y, x) -> grid position (4y, 4x)).  -> bool [B,h*w]"""
    next_conf, s = _f(next_conf), _f(s_d2d)
    B = next_conf.shape[0]
    h, w = hw
    z = np.zeros((B, h * w), np.int64)
    num = nms_select(next_conf, z, z, hw, hw, nms_window=window_size, test_thr=-np.inf, double_check=False)["keep"].sum(1)
    s = s.reshape(B, -1)
    dw = w // 4
    keep = np.zeros((B, h * w), bool)
    for b in range(B):
        top = np.argsort(-s[b], kind="stable")[:min(s.shape[1], int(num[b]))]
        keep[b, (top // dw * 4) * (dw * 4) + top % dw * 4] = True
    return keep


def conv_soft_argmax_mask(next_conf, hw, window_size, stride=1, temperature=1.0):
    """PostProcess 'softargmax_nms' (post_processing.py:93-110) before the threshold.  PARITY UNPINNED: the method is
    kornia.geometry.ConvSoftArgmax2d (kornia 0.6.2, kornia/geometry/subpix/spatial_soft_argmax.py: conv_soft_argmax2d with
    normalized_coordinates=False, eps 1e-8), absent from /root/reference and from this image; restated from the published algorithm:
      e = exp((x - max x) / T); per window (zero padded): den = sum e + eps; offset = sum(e * g) / den with g the window grid in
      kornia's normalised units (-1 .. 1 across the window, x then y); position = offset + window centre in pixels (even windows: the
      mean of the central pixels); the reference rounds it, forms channel0 * w0c + channel1 (:104) and scatters True there."""
    x = _f(next_conf).astype(np.float64)
    B = x.shape[0]
    h, w = hw
    ws = window_size
    pad = ws // 2 if stride == 1 else 0
    x = x.reshape(B, h, w)
    e = np.exp(((x - x.max(axis=(1, 2), keepdims=True)) / temperature).astype(np.float32)).astype(np.float64)
    ep = np.pad(e, ((0, 0), (pad, pad), (pad, pad)))
    gy, gx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    gxp, gyp = np.pad(gx, pad), np.pad(gy, pad)
    ho, wo = (h + 2 * pad - ws) // stride + 1, (w + 2 * pad - ws) // stride + 1
    off = np.linspace(-1.0, 1.0, ws) if ws > 1 else np.zeros(1)
    c1, c2 = (ws // 2, ws // 2 + 1) if ws % 2 else (ws // 2 - 1, ws // 2 + 1)
    keep = np.zeros((B, h * w), bool)
    for oy in range(ho):
        for ox in range(wo):
            win = ep[:, oy * stride:oy * stride + ws, ox * stride:ox * stride + ws]
            den = win.sum(axis=(1, 2)) + 1e-8
            cx = (win * off[None, None, :]).sum(axis=(1, 2)) / den + gxp[oy * stride + c1:oy * stride + c2, ox * stride + c1:ox * stride + c2].mean()
            cy = (win * off[None, :, None]).sum(axis=(1, 2)) / den + gyp[oy * stride + c1:oy * stride + c2, ox * stride + c1:ox * stride + c2].mean()
            flat = np.rint(cx.astype(np.float32)).astype(np.int64) * w + np.rint(cy.astype(np.float32)).astype(np.int64)
            if (flat < 0).any() or (flat >= h * w).any():
                raise IndexError("softargmax_nms: index out of range (torch.scatter fails in the reference too)")
            keep[np.arange(B), flat] = True
    return keep


def nms_select(next_conf01, next_idx01, next_idx10, hw0, hw1, nms_window=5, test_thr=0.2, pre=(), border_rm=0,
               valid_hw=None, double_check=True, extra_keep=None):
    """pre: sequence of (pre_conf [B,hp*wp], (hp,wp), pre_thr)."""
    next_conf01, next_idx01, next_idx10 = _f(next_conf01), _i(next_idx01), _i(next_idx10)
    B, N = next_conf01.shape
    vh = None if valid_hw is None else np.ascontiguousarray(valid_hw, dtype=np.int32)
    pre = list(pre) + [(None, (1, 1), 0.0)] * (2 - len(pre))
    n_pre = sum(p[0] is not None for p in pre)
    pc0, pc1 = _f(pre[0][0]), _f(pre[1][0])
    keep = np.empty((B, N), np.uint8)
    bi, ii, ji = (np.empty(B * N, np.int64) for _ in range(3))
    mc = np.empty(B * N, np.float32)
    n = lib().orc_nms_select(_p(next_conf01), _p(next_idx01), _p(next_idx10), C.c_int(nms_window), C.c_float(test_thr),
                             C.c_int(n_pre), _p(pc0), *_ci(*pre[0][1]), C.c_float(pre[0][2]),
                             _p(pc1), *_ci(*pre[1][1]), C.c_float(pre[1][2]), C.c_int(border_rm), _p(vh),
                             C.c_int(int(double_check)), _p(keep), _p(bi), _p(ii), _p(ji), _p(mc),
                             *_ci(B, hw0[0], hw0[1], hw1[0], hw1[1]), _p(_u8(extra_keep)))
    return dict(keep=keep.astype(bool), b_ids=bi[:n].copy(), i_ids=ii[:n].copy(), j_ids=ji[:n].copy(), mconf=mc[:n].copy())


def window_warp_idx(idx, H, W, ws=5):
    idx = _i(idx)
    B, N = idx.shape
    out = np.empty((B, N, ws * ws, 2), np.int64)
    lib().orc_window_warp_idx(_p(idx), _p(out), *_ci(B, N, H, W, ws))
    return out


# ------------------------------------------------------------------------------- §8(f)-1: the callers, token-major
def linear(x, w, bias=None):
    """y[..., n] = chain_k fmaf(x[..., k], w[n, k]) (+ bias[n]): nn.Conv2d(dim, dim, 1) on tokens / nn.Linear
    (src/model/modules/quadtree_attention.py:79-81,98)."""
    x, w, bias = _f(x), _f(w), _f(bias)
    N, K = w.shape[0], int(np.prod(w.shape[1:]))
    M = x.size // K
    y = np.empty(x.shape[:-1] + (N,), np.float32)
    lib().orc_linear(_p(x), _p(w), _p(bias), _p(y), C.c_int64(M), *_ci(N, K))
    return y


def token_pool(x, H, W):
    """avg_pool2d(kernel 2, stride 2) of a token-major [B,H*W,C] tensor -> [B,(H//2)*(W//2),C] (:86-90)."""
    x = _f(x)
    B, _, Cc = x.shape
    out = np.empty((B, (H // 2) * (W // 2), Cc), np.float32)
    lib().orc_token_pool(_p(x), _p(out), *_ci(B, H, W, Cc))
    return out


def dwconv3x3_tokens(x, w, bias, H, W, pre_relu=False, post_gelu=False, add_input=False):
    """DWConv of Mlp / PosCNN on token-major [B,H*W,C] (transformer.py:52-94, gvt.py:397-411); w [C,1,3,3] or [C,9]."""
    x, w = _f(x), _f(np.asarray(w).reshape(-1, 9))
    B, _, Cc = x.shape
    bias = None if bias is None else _f(bias)
    out = np.empty_like(x)
    lib().orc_dwconv3x3_tokens(_p(x), _p(w), _p(bias) if bias is not None else None, _p(out), *_ci(B, H, W, Cc),
                               C.c_int(int(pre_relu) | 2 * int(post_gelu) | 4 * int(add_input)))
    return out


def layer_norm(x, gamma, beta, eps=1e-5, residual=None):
    """nn.LayerNorm over the last axis (+ residual), double accumulation."""
    x, gamma, beta = _f(x), _f(gamma), _f(beta)
    residual = None if residual is None else _f(residual)
    out = np.empty_like(x)
    Cc = x.shape[-1]
    lib().orc_layer_norm(_p(x), _p(gamma), _p(beta), _p(residual) if residual is not None else None, _p(out),
                         C.c_int64(x.size // Cc), C.c_int(Cc), C.c_float(eps))
    return out


def window_attn(qkv, H, W, nhead, ws, scale):
    """GroupAttention.forward_mask for the real tokens (cascade_attention.py:124-157): qkv [B,H*W,3*C] -> [B,H*W,C]."""
    qkv = _f(qkv)
    B, HW, C3 = qkv.shape
    Cc = C3 // 3
    assert ws * ws <= 64 and Cc // nhead <= 64
    out = np.empty((B, HW, Cc), np.float32)
    lib().orc_window_attn(_p(qkv), _p(out), *_ci(B, H, W, nhead, Cc // nhead, ws), C.c_float(scale))
    return out


def _nchw(t, h, w):
    B, _, Cc = t.shape
    return np.ascontiguousarray(t.reshape(B, h, w, Cc).transpose(0, 3, 1, 2))


def quadtree_attention_block(x, target, hw, hw1, wq, wk, wv, level_weight, wp, bp, nhead, topks, scale=3,
                             bq=None, bk=None, bv=None):
    """QuadtreeAttention.forward with attn_type 'B' (src/model/modules/quadtree_attention.py:68-100):
    x [B,N,C] / target [B,N1,C] tokens -> ([B,N,C], QTAttB per-level dicts)."""
    q, k, v = linear(x, wq, bq), linear(target, wk, bk), linear(target, wv, bv)
    (h, w), (h1, w1) = hw, hw1
    qs, ks, vs = [], [], []
    for i in range(scale):
        qs.append(_nchw(q, h, w)); ks.append(_nchw(k, h1, w1)); vs.append(_nchw(v, h1, w1))
        if i != scale - 1:
            q, k, v = token_pool(q, h, w), token_pool(k, h1, w1), token_pool(v, h1, w1)
            h, w, h1, w1 = h // 2, w // 2, h1 // 2, w1 // 2
    msg, levels = qtattb_forward(qs, ks, vs, level_weight, nhead, topks)
    return linear(msg.reshape(x.shape[0], -1, x.shape[2]), wp, bp), levels


def cascade_quadtree_attention_block(x, target, hw, hw1, idx, wq, wk, wv, wp, bp, nhead, dilated=1, rel_pos=None):
    """CascadeQuadtreeAttention.forward (:152-176) -> (x [B,N,C], upsampled_idx [B,N,4*KW])."""
    q, k, v = linear(x, wq), linear(target, wk), linear(target, wv)
    msg, up = cascade_attn(q, k, v, idx, hw, hw1, nhead, dilated, rel_pos)
    return linear(msg, wp, bp), up
