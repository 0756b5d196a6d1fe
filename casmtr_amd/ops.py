"""Tensor-level front end of the C ABI: argument checks, output allocation, current HIP stream + device guard.

PyTorch is plumbing here (device memory, streams); every computation below is a hand-written gfx950 kernel in
libcasmtr_hip.so.  The preconditions mirror the reference's TORCH_CHECKs (score_computation.cpp:6-8): tensors must
be device-resident and contiguous, fp32 values, int64 indices -- violations raise RuntimeError, nothing falls back
to the CPU.
"""
from __future__ import annotations

import os

import torch

from . import _lib


def _chk(t, name, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA (HIP) tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _u8(mask):
    """bool / uint8 mask -> contiguous uint8 (same bytes; no copy for contiguous bool tensors)."""
    if mask is None:
        return None
    if mask.dtype == torch.bool:
        mask = mask.contiguous().view(torch.uint8)
    return _chk(mask.contiguous(), "mask", torch.uint8)


# ------------------------------------------------------------------------------------------------ drop-in primitives
def qta_score_fwd(query, key, index):
    _chk(query, "query"), _chk(key, "key"), _chk(index, "index", torch.int64)
    B, N1, four, H, D = query.shape
    assert four == 4
    N2, K = key.shape[1], index.shape[2]
    out = torch.empty((B, N1, 4, K, H), device=query.device, dtype=torch.float32)
    with torch.cuda.device(query.device):
        _lib.check(_lib.lib().casmtr_qta_score_fwd(_ptr(query), _ptr(key), _ptr(index), _ptr(out), B, N1, N2, K, H, D,
                                                    _stream()), "qta_score_fwd")
    return out


def qta_score_bwd(grad, query, key, index):
    _chk(grad, "grad"), _chk(query, "query"), _chk(key, "key"), _chk(index, "index", torch.int64)
    B, N1, _, H, D = query.shape
    N2, K = key.shape[1], index.shape[2]
    dq, dk = torch.empty_like(query), torch.empty_like(key)
    with torch.cuda.device(query.device):
        _lib.check(_lib.lib().casmtr_qta_score_bwd(_ptr(grad), _ptr(query), _ptr(key), _ptr(index), _ptr(dq), _ptr(dk),
                                                    B, N1, N2, K, H, D, _stream()), "qta_score_bwd")
    return dq, dk


def qta_value_agg_fwd(score, value, index, output):
    _chk(score, "score"), _chk(value, "value"), _chk(index, "index", torch.int64), _chk(output, "output")
    B, N, K, H = score.shape
    M, D = value.shape[1], value.shape[3]
    with torch.cuda.device(score.device):
        _lib.check(_lib.lib().casmtr_qta_value_agg_fwd(_ptr(score), _ptr(value), _ptr(index), _ptr(output), B, N, K, H,
                                                        M, D, _stream()), "qta_value_agg_fwd")


def qta_value_agg_bwd(grad_out, score, value, index, grad_score, grad_value):
    for t, n in ((grad_out, "grad_out"), (score, "score"), (value, "value"), (grad_score, "grad_score"),
                 (grad_value, "grad_value")):
        _chk(t, n)
    _chk(index, "index", torch.int64)
    B, N, K, H = score.shape
    M, D = value.shape[1], value.shape[3]
    with torch.cuda.device(score.device):
        _lib.check(_lib.lib().casmtr_qta_value_agg_bwd(_ptr(grad_out), _ptr(score), _ptr(value), _ptr(index),
                                                        _ptr(grad_score), _ptr(grad_value), B, N, K, H, M, D, _stream()),
                   "qta_value_agg_bwd")


def window_score_fwd(query, key, index):
    _chk(query, "query"), _chk(key, "key"), _chk(index, "index", torch.int64)
    B, N1, Cc = query.shape
    N2, K = key.shape[1], index.shape[2]
    out = torch.empty((B, N1, K), device=query.device, dtype=torch.float32)
    with torch.cuda.device(query.device):
        _lib.check(_lib.lib().casmtr_window_score_fwd(_ptr(query), _ptr(key), _ptr(index), _ptr(out), B, N1, N2, K, Cc,
                                                       _stream()), "window_score_fwd")
    return out


def window_score_bwd(grad, query, key, index):
    _chk(grad, "grad"), _chk(query, "query"), _chk(key, "key"), _chk(index, "index", torch.int64)
    B, N1, Cc = query.shape
    N2, K = key.shape[1], index.shape[2]
    dq, dk = torch.empty_like(query), torch.empty_like(key)
    with torch.cuda.device(query.device):
        _lib.check(_lib.lib().casmtr_window_score_bwd(_ptr(grad), _ptr(query), _ptr(key), _ptr(index), _ptr(dq),
                                                       _ptr(dk), B, N1, N2, K, Cc, _stream()), "window_score_bwd")
    return dq, dk


# ------------------------------------------------------------------------------------------------ fused kernels
def nchw_to_tokens(x):
    """[B,C,h,w] -> [B,h*w,C]"""
    _chk(x, "x")
    B, Cc, h, w = x.shape
    out = torch.empty((B, h * w, Cc), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().casmtr_nchw_to_tokens(_ptr(x), _ptr(out), B, Cc, h * w, _stream()), "nchw_to_tokens")
    return out


def _is_channels_last(x):
    B, C, h, w = x.shape
    return x.dtype == torch.float32 and x.stride() == (h * w * C, 1, w * C, C)


def nchw_to_tokens_multi(xs):
    """list of [B,C_i,h_i,w_i] (same B) -> list of [B,h_i*w_i,C_i], one launch for up to 9 tensors.
    Tensors that already are channels_last in memory (what MIOpen convolutions prefer) are token-major as they stand: they
    are returned as zero-copy views and skip the kernel."""
    import ctypes as C
    outs = [None] * len(xs)
    todo = []
    for i, x in enumerate(xs):
        if x.dim() == 4 and x.is_cuda and _is_channels_last(x):
            outs[i] = x.permute(0, 2, 3, 1).reshape(x.shape[0], x.shape[2] * x.shape[3], x.shape[1])
        else:
            todo.append(i)
    for j in range(0, len(todo), 9):
        ids = todo[j:j + 9]
        chunk = [xs[i] if xs[i].is_contiguous() else xs[i].contiguous() for i in ids]
        for x in chunk:
            _chk(x, "x")
        B = chunk[0].shape[0]
        if any(x.shape[0] != B for x in chunk):
            raise RuntimeError("nchw_to_tokens_multi: tensors must share the batch size")
        res = [torch.empty((B, x.shape[2] * x.shape[3], x.shape[1]), device=x.device, dtype=torch.float32) for x in chunk]
        n = len(chunk)
        src = (C.c_void_p * n)(*[x.data_ptr() for x in chunk])
        dst = (C.c_void_p * n)(*[r.data_ptr() for r in res])
        cs = (C.c_int * n)(*[x.shape[1] for x in chunk])
        hws = (C.c_int * n)(*[x.shape[2] * x.shape[3] for x in chunk])
        with torch.cuda.device(chunk[0].device):
            _lib.check(_lib.lib().casmtr_nchw_to_tokens_multi(C.cast(src, C.c_void_p), C.cast(dst, C.c_void_p),
                                                              C.cast(cs, C.c_void_p), C.cast(hws, C.c_void_p), n, B,
                                                              _stream()), "nchw_to_tokens_multi")
        for i, r in zip(ids, res):
            outs[i] = r
    return outs


_LINEAR_GEMM = [None]   # process-wide default of linear_multi / linear_quads_multi (None: CASMTR_LINEAR_GEMM or "exact")


def linear_gemm_mode(requested=None):
    """How the projections multiply.  "exact" (default): casmtr_linear[_quads]_fwd, the fp32 MFMA fmaf chain (v_mfma_f32_32x32x2_f32,
    k ascending) -- the oracle's arithmetic.  "split": casmtr_linear_split_fwd, fp32-accurate on the f16 matrix pipe (two-term f16
    split of the power-of-two-normalised operands, 3 MFMA products, fp32 accumulate; |y - y_exact| <= 2^-15 |x_m||w_n|, measured ~1e-7).
    The throughput paths opt in (pipeline.HotPath(callers), model.timing); CASMTR_LINEAR_GEMM overrides both."""
    import os
    mode = os.environ.get("CASMTR_LINEAR_GEMM") or requested or _LINEAR_GEMM[0] or "exact"
    if mode not in ("exact", "split"):
        raise ValueError(f"linear gemm mode {mode!r} (exact | split)")
    return mode


def prepare_split_weight(w):
    """weight [N, K] (fp32, contiguous, device-resident) -> the prepared form casmtr_linear_split_fwd multiplies with (f16 split tile image
    + per-row factors), or None when the split kernel does not cover the shape.  Depends on the weight's VALUES: callers that keep it
    (modules/quadtree_block.py caches it per parameter, keyed by the parameter's data pointer and version counter) must re-prepare after
    the weight changes.  ops.linear_multi(gemm="split") without `preps` prepares on every call (a [N]-block kernel, microseconds)."""
    w = _chk(w.reshape(w.shape[0], -1), "w")
    N, K = w.shape
    l = _lib.lib()
    nbytes = l.casmtr_linear_split_prep_bytes(N, K)
    if nbytes == 0:
        return None
    prep = torch.empty(nbytes, device=w.device, dtype=torch.uint8)
    with torch.cuda.device(w.device):
        _lib.check(l.casmtr_linear_split_prep(_ptr(w), _ptr(prep), N, K, _stream()), "linear_split_prep")
    return prep


def _linear_split(xs, ws, bs, ys, M, N, K, h, w, preps=None):
    """-> False when the split path does not cover the shape (caller runs the exact kernel)"""
    import ctypes as C
    if N % 128 or K % 32 or K > 256:
        return False
    preps = [prepare_split_weight(wt) for wt in ws] if preps is None else list(preps)
    if any(p is None for p in preps):
        return False
    n = len(xs)
    arr = lambda ts: C.cast((C.c_void_p * n)(*[_ptr(t) for t in ts]), C.c_void_p)
    with torch.cuda.device(xs[0].device):
        _lib.check(_lib.lib().casmtr_linear_split_fwd(arr(xs), arr(preps), arr(bs), arr(ys), n, M, N, K, h, w, _stream()),
                   "linear_split_fwd")
    return True


def linear_multi(xs, ws, biases=None, gemm=None, preps=None):
    """y_i = x_i @ w_i^T (+ b_i) for up to 4 problems of one shape in one launch (k-ascending fp32 MFMA chain; gemm="split": see
    linear_gemm_mode).  x_i [..., K] token-major, w_i [N, K] (a [N,K,1,1] conv weight is viewed), b_i [N] or None."""
    import ctypes as C
    n = len(xs)
    biases = [None] * n if biases is None else list(biases)
    xs = [_chk(x, "x") for x in xs]
    ws = [_chk(w.reshape(w.shape[0], -1), "w") for w in ws]
    bs = [_chk(b, "bias") for b in biases]
    N, K = ws[0].shape
    M = xs[0].numel() // K
    if any(x.shape[-1] != K or x.numel() != M * K for x in xs) or any(tuple(w.shape) != (N, K) for w in ws):
        raise RuntimeError("linear_multi: all problems must share (M, N, K)")
    ys = [torch.empty(x.shape[:-1] + (N,), device=x.device, dtype=torch.float32) for x in xs]
    if linear_gemm_mode(gemm) == "split" and _linear_split(xs, ws, bs, ys, M, N, K, 0, 0, preps):
        return ys
    arr = lambda ts: C.cast((C.c_void_p * n)(*[_ptr(t) for t in ts]), C.c_void_p)
    with torch.cuda.device(xs[0].device):
        _lib.check(_lib.lib().casmtr_linear_fwd(arr(xs), arr(ws), arr(bs), arr(ys), n, M, N, K, _stream()), "linear_fwd")
    return ys


def linear(x, w, bias=None, gemm=None, prep=None):
    return linear_multi([x], [w], [bias], gemm=gemm, preps=None if prep is None else [prep])[0]


def linear_quads_multi(xs, ws, biases, h, w, gemm=None, preps=None):
    """linear_multi with the results written quad-major per head: x_i [B, h*w, K] -> [B, N/32, (h/2)*(w/2), 4, 32] (the layout
    tokens_to_quads produces), one launch, no layout pass.  h, w even, N % 32 == 0."""
    import ctypes as C
    n = len(xs)
    biases = [None] * n if biases is None else list(biases)
    xs = [_chk(x, "x") for x in xs]
    ws = [_chk(wt.reshape(wt.shape[0], -1), "w") for wt in ws]
    bs = [_chk(b, "bias") for b in biases]
    N, K = ws[0].shape
    B = xs[0].shape[0]
    M = B * h * w
    if any(tuple(x.shape) != (B, h * w, K) for x in xs) or any(tuple(wt.shape) != (N, K) for wt in ws):
        raise RuntimeError("linear_quads_multi: x_i must be [B, h*w, K] and all problems share (N, K)")
    ys = [torch.empty((B, N // 32, (h // 2) * (w // 2), 4, 32), device=x.device, dtype=torch.float32) for x in xs]
    if linear_gemm_mode(gemm) == "split" and h % 2 == 0 and w % 2 == 0 and _linear_split(xs, ws, bs, ys, M, N, K, h, w, preps):
        return ys
    arr = lambda ts: C.cast((C.c_void_p * n)(*[_ptr(t) for t in ts]), C.c_void_p)
    with torch.cuda.device(xs[0].device):
        _lib.check(_lib.lib().casmtr_linear_quads_fwd(arr(xs), arr(ws), arr(bs), arr(ys), n, M, N, K, h, w, _stream()), "linear_quads_fwd")
    return ys


def linear_quads_pyramid_multi(xs, ws, biases, h, w, levels, preps=None, outs=None):
    """The q / k / v projections of QuadtreeAttention.forward together with their avg_pool2d pyramid in ONE launch
    (casmtr_linear_split_pyramid_fwd, csrc/linear_pc.hip; split-f16 GEMM only).  x_i [B, h*w, K] token-major, w_i [N, K], up to 8
    problems (both directions of a layer).
    -> per problem a list of `levels` tensors, finest first: quad-major [B, N/32, (h/2^(l+1))*(w/2^(l+1)), 4, 32] for every level but
    the last, the last (coarsest) one token-major [B, (h/2^l)*(w/2^l), N] (levels == 1: the quad-major projection alone) -- what
    ops.linear_quads_multi followed by ops.quad_pool_multi(.., to_tokens=last) returns, bit for bit.  outs: the same structure
    preallocated by the caller (e.g. slices of doubled-batch operands); levels in (1, 2, 3); None when the kernel does not cover the
    shape (the caller runs the separate launches)."""
    import ctypes as C
    n = len(xs)
    biases = [None] * n if biases is None else list(biases)
    xs = [_chk(x, "x") for x in xs]
    ws = [_chk(wt.reshape(wt.shape[0], -1), "w") for wt in ws]
    bs = [_chk(b, "bias") for b in biases]
    N, K = ws[0].shape
    B = xs[0].shape[0]
    if any(tuple(x.shape) != (B, h * w, K) for x in xs) or any(tuple(wt.shape) != (N, K) for wt in ws):
        raise RuntimeError("linear_quads_pyramid_multi: x_i must be [B, h*w, K] and all problems share (N, K)")
    need = max(2, 1 << (levels - 1))   # every level but the coarsest is stored as quads
    if (levels not in (1, 2, 3) or K not in (128, 256) or N % 128 or (K == 256 and N % 256) or n > 8 or n * N > 2048 or h % need or w % need):
        return None
    preps = [prepare_split_weight(wt) for wt in ws] if preps is None else list(preps)
    if any(p is None for p in preps):
        return None
    dev = xs[0].device
    shapes = [(B, N // 32, (h // 2) * (w // 2), 4, 32)]
    if levels == 2:
        shapes.append((B, (h // 2) * (w // 2), N))
    elif levels == 3:
        shapes += [(B, N // 32, (h // 4) * (w // 4), 4, 32), (B, (h // 4) * (w // 4), N)]
    if outs is None:
        outs = [[torch.empty(sh, device=dev, dtype=torch.float32) for sh in shapes] for _ in xs]
    elif (len(outs) != n or any(len(o) != levels for o in outs)
          or any(tuple(t.shape) != sh or not t.is_contiguous() or t.dtype != torch.float32 or t.device != dev for o in outs for t, sh in zip(o, shapes))):
        raise RuntimeError("linear_quads_pyramid_multi: outs must hold contiguous fp32 tensors of the level shapes")
    arr = lambda ts: C.cast((C.c_void_p * n)(*[_ptr(t) for t in ts]), C.c_void_p)
    lvl = lambda l: arr([o[l] for o in outs]) if l < levels else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().casmtr_linear_split_pyramid_fwd(arr(xs), arr(preps), arr(bs), lvl(0), lvl(1), lvl(2), int(levels == 2),
                                                              n, B, h, w, N, K, _stream()), "linear_split_pyramid_fwd")
    return [list(o) for o in outs]


def quad_pool_multi(xs, h, w, to_tokens=False):
    """avg_pool2d(2, 2) on quad-major tensors [B, H, (h/2)*(w/2), 4, 32] (h x w tokens) -> the pooled level quad-major
    [B, H, (h/4)*(w/4), 4, 32], or token-major [B, (h/2)*(w/2), H*32] with to_tokens; one launch."""
    import ctypes as C
    n = len(xs)
    xs = [_chk(x, "x") for x in xs]
    B, H = xs[0].shape[:2]
    if any(tuple(x.shape) != (B, H, (h // 2) * (w // 2), 4, 32) for x in xs):
        raise RuntimeError("quad_pool_multi: tensors must be quad-major [B, H, (h/2)*(w/2), 4, 32]")
    shape = (B, (h // 2) * (w // 2), H * 32) if to_tokens else (B, H, (h // 4) * (w // 4), 4, 32)
    ys = [torch.empty(shape, device=x.device, dtype=torch.float32) for x in xs]
    arr = lambda ts: C.cast((C.c_void_p * n)(*[_ptr(t) for t in ts]), C.c_void_p)
    with torch.cuda.device(xs[0].device):
        _lib.check(_lib.lib().casmtr_quad_pool_fwd(arr(xs), arr(ys), n, B, h, w, H * 32, int(bool(to_tokens)), _stream()), "quad_pool_fwd")
    return ys


def token_pool_multi(xs, H, W):
    """avg_pool2d(2, 2) on token-major tensors: list of [B,H*W,C] -> list of [B,(H//2)*(W//2),C], one launch."""
    import ctypes as C
    n = len(xs)
    xs = [_chk(x, "x") for x in xs]
    B, HW, Cc = xs[0].shape
    if HW != H * W or any(tuple(x.shape) != (B, HW, Cc) for x in xs):
        raise RuntimeError("token_pool_multi: tensors must share the shape [B, H*W, C]")
    ys = [torch.empty((B, (H // 2) * (W // 2), Cc), device=x.device, dtype=torch.float32) for x in xs]
    arr = lambda ts: C.cast((C.c_void_p * n)(*[_ptr(t) for t in ts]), C.c_void_p)
    with torch.cuda.device(xs[0].device):
        _lib.check(_lib.lib().casmtr_token_pool_fwd(arr(xs), arr(ys), n, B, H, W, Cc, _stream()), "token_pool_fwd")
    return ys


def dwconv3x3_tokens(x, weight, bias, H, W, pre_relu=False, post_gelu=False, add_input=False):
    """depth-wise 3x3 (stride 1, zero pad 1) on token-major x [B,H*W,C]; weight [C,1,3,3]; optional fused ReLU-in / GELU-out /
    + input (Mlp's DWConv, PosCNN)."""
    _chk(x, "x"), _chk(weight, "weight"), _chk(bias, "bias")
    B, HW, Cc = x.shape
    if HW != H * W or weight.numel() != 9 * Cc:
        raise RuntimeError("dwconv3x3_tokens: x must be [B, H*W, C] and weight [C, 1, 3, 3]")
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().casmtr_dwconv3x3_tokens_fwd(_ptr(x), _ptr(weight), _ptr(bias), _ptr(y), B, H, W, Cc,
                                                          int(pre_relu) | 2 * int(post_gelu) | 4 * int(add_input), _stream()),
                   "dwconv3x3_tokens_fwd")
    return y


def layer_norm(x, gamma, beta, eps=1e-5, residual=None):
    """nn.LayerNorm over the last axis of a contiguous fp32 tensor, optionally + residual (same shape) after the affine map."""
    _chk(x, "x"), _chk(gamma, "gamma"), _chk(beta, "beta"), _chk(residual, "residual")
    Cc = x.shape[-1]
    if residual is not None and residual.shape != x.shape:
        raise RuntimeError("layer_norm: residual must have x's shape")
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().casmtr_layer_norm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(residual), _ptr(y), x.numel() // Cc, Cc,
                                                    float(eps), _stream()), "layer_norm_fwd")
    return y


def window_attn(qkv, H, W, nhead, ws, scale):
    """window self-attention on the fused projection qkv [B,H*W,3*C] of un-padded tokens -> [B,H*W,C] (head_dim 32, ws 7)."""
    _chk(qkv, "qkv")
    B, HW, C3 = qkv.shape
    Cc = C3 // 3
    if HW != H * W or C3 != 3 * Cc or Cc % nhead:
        raise RuntimeError("window_attn: qkv must be [B, H*W, 3*C]")
    y = torch.empty((B, HW, Cc), device=qkv.device, dtype=torch.float32)
    with torch.cuda.device(qkv.device):
        _lib.check(_lib.lib().casmtr_window_attn_fwd(_ptr(qkv), _ptr(y), B, H, W, nhead, Cc // nhead, ws, float(scale), _stream()),
                   "window_attn_fwd")
    return y


def pola_attn(q, k0, v0, bias_table, H, W, nhead, ws, scale):
    """POLA neighbourhood attention (3 x 3 windows of ws x ws around each query window, relative position bias) on un-padded
    projected tokens q, k0, v0 [B,H*W,C] (k0, v0 bias-free) -> [B,H*W,C]; head_dim 32, ws 7."""
    _chk(q, "q"), _chk(k0, "k0"), _chk(v0, "v0"), _chk(bias_table, "bias_table")
    B, HW, Cc = q.shape
    if HW != H * W or k0.shape != q.shape or v0.shape != q.shape or tuple(bias_table.shape) != ((4 * ws - 1) ** 2, nhead):
        raise RuntimeError("pola_attn: q, k0, v0 must be [B, H*W, C] and bias_table [(4 ws - 1)^2, nhead]")
    y = torch.empty_like(q)
    with torch.cuda.device(q.device):
        _lib.check(_lib.lib().casmtr_pola_attn_fwd(_ptr(q), _ptr(k0), _ptr(v0), _ptr(bias_table), _ptr(y), B, H, W, nhead, Cc // nhead, ws,
                                                   float(scale), _stream()), "pola_attn_fwd")
    return y


def qta_coarse_level(q, k, v, nhead, topk, w_level=None, want_message=True, want_tab=False, want_topk=True):
    """q [B,L,C], k/v [B,S,C] tokens -> dict(message, acc, topk_score, topk_idx, topk_tab, probs_ws).
    want_tab: also the compact per-head table [B,H,L,topk] int32 that qta_fine_level_quad takes as `parents`.
    want_topk=False: the reference's [B,L,topk,H] score / int64 index tensors (:170-175) are not written (the fused module path
    hands the lists on as the int32 table); honoured by the default kernel, the three-kernel path always writes them."""
    _chk(q, "q"), _chk(k, "k"), _chk(v, "v")
    B, L, Cc = q.shape
    S, D = k.shape[1], Cc // nhead
    l = _lib.lib()
    n_ws = l.casmtr_qta_coarse_level_ws_floats_k(B, L, S, nhead, topk)
    want_topk = want_topk or n_ws > 1 or not want_tab
    ws = torch.empty(n_ws, device=q.device, dtype=torch.float32)
    msg = torch.empty((B, L, nhead, D), device=q.device, dtype=torch.float32) if want_message else None
    acc = torch.empty((B, L, nhead, D), device=q.device, dtype=torch.float32) if w_level is not None else None
    tab = torch.empty((B, nhead, L, topk), device=q.device, dtype=torch.int32) if want_tab else None
    while True:
        ts = torch.empty((B, L, topk, nhead), device=q.device, dtype=torch.float32) if want_topk else None
        ti = torch.empty((B, L, topk, nhead), device=q.device, dtype=torch.int64) if want_topk else None
        with torch.cuda.device(q.device):
            rc = l.casmtr_qta_coarse_level_tab_fwd(_ptr(q), _ptr(k), _ptr(v), 1.0 / D ** 0.5, topk,
                                                   0.0 if w_level is None else float(w_level), _ptr(ws), _ptr(msg),
                                                   _ptr(acc), _ptr(ts), _ptr(ti), _ptr(tab), B, L, S, nhead, D, _stream())
        if rc == _lib.ERR_UNSUPPORTED and not want_topk:   # a kernel that cannot skip the int64 lists declined: ask again with them
            want_topk = True
            continue
        _lib.check(rc, "qta_coarse_level_fwd")
        break
    return dict(message=msg, acc=acc, topk_score=ts, topk_idx=ti, topk_tab=tab, probs_ws=ws)


def qta_fine_level(q, key, value, prev_idx, hw0, hw1, nhead, topk, w_level=None, acc_in=None, want_message=True):
    """q [B,h0*w0,C], key/value [B,h1*w1,C], prev_idx [B,L/4,Kp,H] -> dict(message, acc, topk_score, topk_idx)."""
    _chk(q, "q"), _chk(key, "key"), _chk(value, "value"), _chk(prev_idx, "prev_idx", torch.int64), _chk(acc_in, "acc_in")
    B, L, Cc = q.shape
    (h0, w0), (h1, w1) = hw0, hw1
    D, Kp = Cc // nhead, prev_idx.shape[2]
    dev = q.device
    msg = torch.empty((B, L, nhead, D), device=dev, dtype=torch.float32) if want_message else None
    acc = torch.empty((B, L, nhead, D), device=dev, dtype=torch.float32) if w_level is not None else None
    ts = torch.empty((B, L, topk, nhead), device=dev, dtype=torch.float32) if topk > 0 else None
    ti = torch.empty((B, L, topk, nhead), device=dev, dtype=torch.int64) if topk > 0 else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().casmtr_qta_fine_level_fwd(_ptr(q), _ptr(key), _ptr(value), _ptr(prev_idx), 1.0 / D ** 0.5,
                                                         topk, 0.0 if w_level is None else float(w_level), _ptr(acc_in),
                                                         _ptr(msg), _ptr(acc), _ptr(ts), _ptr(ti), B, h0, w0, h1, w1,
                                                         nhead, D, Kp, _stream()), "qta_fine_level_fwd")
    return dict(message=msg, acc=acc, topk_score=ts, topk_idx=ti)


def nchw_to_quads_multi(xs, tokens=None):
    """list of [B,C_i,h_i,w_i] (same B; C_i % 32 == 0, h_i and w_i even) -> list of quad-major per-head tensors
    [B, C_i/32, (h_i/2)*(w_i/2), 4, 32] (include/casmtr_hip.h), one launch for up to 18 tensors.
    tokens: optional list of bools, True = convert that tensor to plain token-major [B, h_i*w_i, C_i] instead (same launch)."""
    import ctypes as C
    outs = []
    tokens = [False] * len(xs) if tokens is None else list(tokens)
    for j in range(0, len(xs), 18):
        chunk = [x if x.is_contiguous() else x.contiguous() for x in xs[j:j + 18]]
        tk = tokens[j:j + 18]
        for x in chunk:
            _chk(x, "x")
        B = chunk[0].shape[0]
        if any(x.shape[0] != B for x in chunk):
            raise RuntimeError("nchw_to_quads_multi: tensors must share the batch size")
        res = [torch.empty((B, x.shape[2] * x.shape[3], x.shape[1]) if t else (B, x.shape[1] // 32, (x.shape[2] // 2) * (x.shape[3] // 2), 4, 32),
                           device=x.device, dtype=torch.float32) for x, t in zip(chunk, tk)]
        n = len(chunk)
        arr = lambda vals, ty: C.cast((ty * n)(*vals), C.c_void_p)
        with torch.cuda.device(chunk[0].device):
            _lib.check(_lib.lib().casmtr_nchw_to_quads_multi(arr([x.data_ptr() for x in chunk], C.c_void_p),
                                                             arr([r.data_ptr() for r in res], C.c_void_p),
                                                             arr([x.shape[1] for x in chunk], C.c_int), arr([x.shape[2] for x in chunk], C.c_int),
                                                             arr([x.shape[3] for x in chunk], C.c_int), arr([int(t) for t in tk], C.c_int),
                                                             n, B, _stream()),
                       "nchw_to_quads_multi")
        outs += res
    return outs


def nchw_to_quads_grouped(groups, tokens=None):
    """groups: list (one entry per OUTPUT) of G same-shaped [B,C,h,w] tensors -> list of outputs [G*B, ...] (quad-major per head, or
    token-major where tokens[i]): source g of output i lands in rows [g*B, (g+1)*B) -- the two directions of an attention layer share one
    layout launch and every later kernel runs once on the doubled batch.  Up to 18 source tensors per launch."""
    import ctypes as C
    tokens = [False] * len(groups) if tokens is None else list(tokens)
    G = len(groups[0])
    if any(len(g) != G for g in groups):
        raise RuntimeError("nchw_to_quads_grouped: every output needs the same number of sources")
    srcs, dsts, outs = [], [], []
    for grp, t in zip(groups, tokens):
        grp = [x if x.is_contiguous() else x.contiguous() for x in grp]
        for x in grp:
            _chk(x, "x")
        B, Cc, h, w = grp[0].shape
        if any(tuple(x.shape) != (B, Cc, h, w) for x in grp):
            raise RuntimeError("nchw_to_quads_grouped: the sources of one output must share their shape")
        out = torch.empty((G * B, h * w, Cc) if t else (G * B, Cc // 32, (h // 2) * (w // 2), 4, 32), device=grp[0].device, dtype=torch.float32)
        outs.append(out)
        per = out[0].numel() * B * 4   # bytes of one source's share
        for g, x in enumerate(grp):
            srcs.append(x)
            dsts.append((out.data_ptr() + g * per, t))
    B = srcs[0].shape[0]
    if any(x.shape[0] != B for x in srcs):
        raise RuntimeError("nchw_to_quads_grouped: tensors must share the batch size")
    for j in range(0, len(srcs), 18):
        cs, cd = srcs[j:j + 18], dsts[j:j + 18]
        n = len(cs)
        arr = lambda vals, ty: C.cast((ty * n)(*vals), C.c_void_p)
        with torch.cuda.device(cs[0].device):
            _lib.check(_lib.lib().casmtr_nchw_to_quads_multi(arr([x.data_ptr() for x in cs], C.c_void_p), arr([d for d, _ in cd], C.c_void_p),
                                                             arr([x.shape[1] for x in cs], C.c_int), arr([x.shape[2] for x in cs], C.c_int),
                                                             arr([x.shape[3] for x in cs], C.c_int), arr([int(t) for _, t in cd], C.c_int),
                                                             n, B, _stream()),
                       "nchw_to_quads_multi")
    return outs


def tokens_to_quads(x, h, w):
    """token-major [B,h*w,C] -> quad-major per head [B,C/32,(h/2)*(w/2),4,32]"""
    _chk(x, "x")
    B, L, Cc = x.shape
    if L != h * w:
        raise RuntimeError("tokens_to_quads: x must be [B, h*w, C]")
    out = torch.empty((B, Cc // 32, (h // 2) * (w // 2), 4, 32), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().casmtr_tokens_to_quads(_ptr(x), _ptr(out), B, Cc, h, w, _stream()), "tokens_to_quads")
    return out


def topk_idx_to_tab(idx):
    """[B,L,K,H] int64 (the reference's topk_idx) -> [B,H,L,K] int32 (the `parents` table of qta_fine_level_quad)"""
    _chk(idx, "idx", torch.int64)
    B, L, K, H = idx.shape
    tab = torch.empty((B, H, L, K), device=idx.device, dtype=torch.int32)
    with torch.cuda.device(idx.device):
        _lib.check(_lib.lib().casmtr_topk_idx_to_tab(_ptr(idx), _ptr(tab), B, L, K, H, _stream()), "topk_idx_to_tab")
    return tab


def fine_quad_supported(nhead, head_dim, hw0, hw1, Kp, topk):
    """shapes the quad-major fine-level kernel covers (csrc/fine_quad.hip); anything else runs the token-major kernels"""
    return (head_dim == 32 and nhead in (1, 2, 4, 8) and all(v % 2 == 0 and v > 0 for v in (*hw0, *hw1)) and 1 <= Kp <= 32
            and topk <= 16 and topk <= 4 * Kp and (hw1[0] // 2) * (hw1[1] // 2) < (1 << 22)
            and (hw1[0] // 2) * (hw1[1] // 2) * (hw1[1] // 2) < (1 << 32))   # the kernel's reciprocal division by w1 / 2


def qta_fine_level_quad(q, key, value, parents, hw0, hw1, nhead, topk, w_level=None, acc_in=None, want_message=True,
                        want_topk=True, want_tab=None):
    """QTAttB.process_fine_level on quad-major operands: q [B,H,Lq0,4,32], key/value [B,H,Lq1,4,32], parents [B,H,Lq0,Kp] int32
    -> dict(message, acc [B,L,H,D] raster; topk_score, topk_idx [B,L,topk,H] (want_topk); topk_tab [B,H,L,topk] int32 (want_tab,
    default: whenever topk > 0))."""
    _chk(q, "q"), _chk(key, "key"), _chk(value, "value"), _chk(parents, "parents", torch.int32), _chk(acc_in, "acc_in")
    (h0, w0), (h1, w1) = hw0, hw1
    B, H, Lq0 = q.shape[:3]
    L, D, Kp = h0 * w0, 32, parents.shape[3]
    if H != nhead or Lq0 * 4 != L or tuple(key.shape[:3]) != (B, H, (h1 // 2) * (w1 // 2)) or tuple(parents.shape[:3]) != (B, H, Lq0):
        raise RuntimeError("qta_fine_level_quad: operands must be quad-major [B,H,Lq,4,32] with parents [B,H,Lq0,Kp]")
    dev = q.device
    want_tab = (topk > 0) if want_tab is None else (want_tab and topk > 0)
    msg = torch.empty((B, L, nhead, D), device=dev, dtype=torch.float32) if want_message else None
    acc = torch.empty((B, L, nhead, D), device=dev, dtype=torch.float32) if w_level is not None else None
    ts = torch.empty((B, L, topk, nhead), device=dev, dtype=torch.float32) if topk > 0 and want_topk else None
    ti = torch.empty((B, L, topk, nhead), device=dev, dtype=torch.int64) if topk > 0 and want_topk else None
    tab = torch.empty((B, nhead, L, topk), device=dev, dtype=torch.int32) if want_tab else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().casmtr_qta_fine_level_quad_fwd(_ptr(q), _ptr(key), _ptr(value), _ptr(parents), 1.0 / D ** 0.5, topk,
                                                              0.0 if w_level is None else float(w_level), _ptr(acc_in), _ptr(msg),
                                                              _ptr(acc), _ptr(tab), _ptr(ts), _ptr(ti), B, h0, w0, h1, w1, nhead, D,
                                                              Kp, _stream()), "qta_fine_level_quad_fwd")
    return dict(message=msg, acc=acc, topk_score=ts, topk_idx=ti, topk_tab=tab)


def cascade_attn(q, key, value, topk_pos, hw0, hw1, nhead, dilated=1, rel_pos=None, want_idx=True):
    """q [B,h0*w0,C], key/value [B,h1*w1,C], topk_pos [B,L/4,KW,2] -> (message [B,L,C], up_idx [B,L,4KW] | None)."""
    _chk(q, "q"), _chk(key, "key"), _chk(value, "value"), _chk(topk_pos, "topk_pos", torch.int64), _chk(rel_pos, "rel_pos")
    B, L, Cc = q.shape
    (h0, w0), (h1, w1) = hw0, hw1
    KW, D = topk_pos.shape[2], Cc // nhead
    msg = torch.empty((B, L, Cc), device=q.device, dtype=torch.float32)
    up = torch.empty((B, L, 4 * KW), device=q.device, dtype=torch.int64) if want_idx else None
    with torch.cuda.device(q.device):
        _lib.check(_lib.lib().casmtr_cascade_attn_fwd(_ptr(q), _ptr(key), _ptr(value), _ptr(topk_pos), _ptr(rel_pos),
                                                       1.0 / D ** 0.5, int(dilated), _ptr(msg), _ptr(up), B, h0, w0, h1,
                                                       w1, nhead, D, KW, _stream()), "cascade_attn_fwd")
    return msg, up


def cascade_quad_supported(nhead, head_dim, hw0, hw1, KW, dilated):
    """shapes the quad-major cascade kernel covers (csrc/cascade_quad.hip): 5 x 5 windows, dilation 1, even grids"""
    return (head_dim == 32 and nhead in (1, 2, 4, 8) and KW == 25 and dilated == 1 and all(v % 2 == 0 for v in (*hw0, *hw1))
            and hw1[0] >= 10 and hw1[1] >= 10 and (hw1[0] // 2) * (hw1[1] // 2) < (1 << 22))


def cascade_attn_quad(q, key, value, topk_pos, hw0, hw1, nhead, rel_pos=None):
    """CascadeQTAttB on quad-major operands: q [B,H,Lq0,4,32], key/value [B,H,Lq1,4,32], topk_pos [B,Lq0,25,2] -> message [B,L,C]"""
    _chk(q, "q"), _chk(key, "key"), _chk(value, "value"), _chk(topk_pos, "topk_pos", torch.int64), _chk(rel_pos, "rel_pos")
    (h0, w0), (h1, w1) = hw0, hw1
    B, H, Lq0 = q.shape[:3]
    KW = topk_pos.shape[2]
    if H != nhead or Lq0 * 4 != h0 * w0 or tuple(key.shape[:3]) != (B, H, (h1 // 2) * (w1 // 2)) or tuple(topk_pos.shape) != (B, Lq0, KW, 2):
        raise RuntimeError("cascade_attn_quad: operands must be quad-major [B,H,Lq,4,32] with topk_pos [B,Lq0,KW,2]")
    msg = torch.empty((B, h0 * w0, nhead * 32), device=q.device, dtype=torch.float32)
    with torch.cuda.device(q.device):
        _lib.check(_lib.lib().casmtr_cascade_attn_quad_fwd(_ptr(q), _ptr(key), _ptr(value), _ptr(topk_pos), _ptr(rel_pos), 1.0 / 32 ** 0.5,
                                                            _ptr(msg), B, h0, w0, h1, w1, nhead, 32, KW, _stream()), "cascade_attn_quad_fwd")
    return msg


def window_warp_idx(idx, H, W, ws=5):
    _chk(idx, "idx", torch.int64)
    B, N = idx.shape
    out = torch.empty((B, N, ws * ws, 2), device=idx.device, dtype=torch.int64)
    with torch.cuda.device(idx.device):
        _lib.check(_lib.lib().casmtr_window_warp_idx(_ptr(idx), _ptr(out), B, N, H, W, ws, _stream()), "window_warp_idx")
    return out


def ds_gemm_mode(requested=None):
    """Which kernel computes CoarseMatching's similarity matrix.
    'exact' (default): fp32 MFMA = the oracle's fmaf chain for every entry of the [L,S] matrix (casmtr_dual_softmax_fwd).
    'split' (opt-in: CoarseMatching(gemm='split'), what the HotPath pipeline / bench.py select): the matrix on the f16 matrix pipe
        (two-term f16 split, fp32-accurate) + exact re-decision of the near-tie ROW / COLUMN ARGMAX of the logits
        (casmtr_dual_softmax_split_fwd).  What is exact there is `next_idx_c01` / `next_idx_c10`; probabilities, confidences and
        therefore the thresholded mutual-max match list are computed from logits that differ by ~1e-6 from the exact path's: a
        confidence within that distance of `thr`, or two confidences of a row / column that close together, can change the list
        (same class of effect as the device exp against libm's in the exact path; tests audit such entries one by one).
    The environment variable CASMTR_DS_GEMM, when set, overrides `requested` (bench legs, tests).  Read per call."""
    m = os.environ.get("CASMTR_DS_GEMM") or requested or "exact"
    if m not in ("split", "exact"):
        raise RuntimeError(f"dual-softmax GEMM mode {m!r}: expected 'split' or 'exact'")
    return m


def dual_softmax(feat0, feat1, hw0, hw1, temperature, thr, border_rm=0, mask0=None, mask1=None, valid_hw=None,
                 recip=True, want_conf=True, gemm=None, want_sim=False):
    """CoarseMatching numerics.  Returns a dict; match lists are capacity-sized, `n` is a device int64 scalar.
    want_conf: conf_matrix [B,L,S] is written.  Without it neither path writes the similarity matrix (pass 2 recomputes the few
    segments that matter, round 6); want_sim=True asks for it (`sim`: tests)."""
    _chk(feat0, "feat0"), _chk(feat1, "feat1")
    mask0, mask1 = _u8(mask0), _u8(mask1)
    _chk(valid_hw, "valid_hw", torch.int32)
    B, L, Cc = feat0.shape
    S = feat1.shape[1]
    dev = feat0.device
    l = _lib.lib()
    split = ds_gemm_mode(gemm) == "split"
    sim = torch.empty((B, L, S), device=dev, dtype=torch.float32)
    ws = torch.empty(l.casmtr_dual_softmax_split_ws_bytes(B, L, S, Cc) if split else l.casmtr_dual_softmax_ws_bytes(B, L, S),
                     device=dev, dtype=torch.uint8)
    fwd = l.casmtr_dual_softmax_split_fwd if split else l.casmtr_dual_softmax_fwd
    ni01 = torch.empty((B, L), device=dev, dtype=torch.int64)
    nc01 = torch.empty((B, L), device=dev, dtype=torch.float32)
    ni10 = torch.empty((B, S), device=dev, dtype=torch.int64)
    nc10 = torch.empty((B, S), device=dev, dtype=torch.float32)
    bi, ii, ji = (torch.empty(B * L, device=dev, dtype=torch.int64) for _ in range(3))
    mc = torch.empty(B * L, device=dev, dtype=torch.float32)
    n = torch.zeros(1, device=dev, dtype=torch.int64)
    with torch.cuda.device(dev):
        _lib.check(fwd(_ptr(feat0), _ptr(feat1), _ptr(mask0), _ptr(mask1), float(temperature),
                       int(bool(recip)), float(thr), int(border_rm), _ptr(valid_hw), hw0[0], hw0[1],
                       hw1[0], hw1[1], 1 if want_conf else (2 if want_sim else 0), _ptr(sim), _ptr(ws), _ptr(ni01),
                       _ptr(nc01), _ptr(ni10), _ptr(nc10), _ptr(bi), _ptr(ii), _ptr(ji), _ptr(mc),
                       _ptr(n), B, L, S, Cc, _stream()), "dual_softmax_fwd")
    return dict(conf_matrix=sim if want_conf else None, sim=sim if (not want_conf and want_sim) else None, next_idx_c01=ni01, next_conf_c01=nc01, next_idx_c10=ni10,
                next_conf_c10=nc10, b_ids=bi, i_ids=ii, j_ids=ji, mconf=mc, n=n)


class WindowIndex:
    """The window lists of a cascade stage in their implicit form: `topk_pos` [B,(h0/2)*(w0/2),KW,2] (row, col) on the
    (h1/2)x(w1/2) grid, as CascadeFeatureTransformer.get_window_warp_idx produces it (transformer.py:416-440).  Equivalent to
    CascadeQTAttB's `upsampled_idx` [B,h0*w0,4*KW] (modules/quadtree_attention.py:419-450), which is 16x larger and identical
    for the 4 children of a quad; window_match() consumes this form directly, materialize() builds the explicit tensor."""

    def __init__(self, topk_pos, hw0, hw1, dilated=1):
        _chk(topk_pos, "topk_pos", torch.int64)
        self.topk_pos, self.hw0, self.hw1, self.dilated = topk_pos, tuple(int(x) for x in hw0), tuple(int(x) for x in hw1), int(dilated)
        B, Lq, KW, two = topk_pos.shape
        if two != 2 or Lq * 4 != self.hw0[0] * self.hw0[1]:
            raise RuntimeError("WindowIndex: topk_pos must be [B,(h0/2)*(w0/2),KW,2]")
        self._full = None

    @property
    def shape(self):
        B, Lq, KW, _ = self.topk_pos.shape
        return (B, Lq * 4, 4 * KW)

    def materialize(self):
        if self._full is None:
            B, N, K = self.shape
            out = torch.empty((B, N, K), device=self.topk_pos.device, dtype=torch.int64)
            with torch.cuda.device(out.device):
                _lib.check(_lib.lib().casmtr_window_expand_idx(_ptr(self.topk_pos), _ptr(out), B, self.hw0[0], self.hw0[1],
                                                               self.hw1[0], self.hw1[1], K // 4, self.dilated, _stream()),
                           "window_expand_idx")
            self._full = out
        return self._full


def window_match(feat_q, feat_k, idx, temperature=1.0, mask_q=None, mask_k=None, recip=True, want_conf=True, hw=None):
    """idx: int64 [B,N,K] (the reference's upsampled_idx) or a WindowIndex (implicit form, no index tensor in HBM)."""
    _chk(feat_q, "feat_q"), _chk(feat_k, "feat_k")
    mask_q, mask_k = _u8(mask_q), _u8(mask_k)
    B, N, Cc = feat_q.shape
    M, K = feat_k.shape[1], idx.shape[2]
    dev = feat_q.device
    conf = torch.empty((B, N, K), device=dev, dtype=torch.float32) if want_conf else None
    nc = torch.empty((B, N), device=dev, dtype=torch.float32)
    ni = torch.empty((B, N), device=dev, dtype=torch.int64)
    if isinstance(idx, WindowIndex):
        if tuple(idx.shape[:2]) != (B, N) or idx.hw1[0] * idx.hw1[1] != M:
            raise RuntimeError("window_match: WindowIndex does not match the feature shapes")
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().casmtr_window_match_pos_fwd(_ptr(feat_q), _ptr(feat_k), _ptr(idx.topk_pos), _ptr(mask_q),
                                                               _ptr(mask_k), float(temperature), int(bool(recip)), idx.dilated,
                                                               _ptr(conf), _ptr(nc), _ptr(ni), B, idx.hw0[0], idx.hw0[1],
                                                               idx.hw1[0], idx.hw1[1], K // 4, Cc, _stream()),
                       "window_match_pos_fwd")
        return dict(conf_matrix=conf, next_conf=nc, next_idx=ni)
    _chk(idx, "idx", torch.int64)
    h, w = hw if hw is not None else (0, 0)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().casmtr_window_match_fwd(_ptr(feat_q), _ptr(feat_k), _ptr(idx), _ptr(mask_q), _ptr(mask_k),
                                                       float(temperature), int(bool(recip)), _ptr(conf), _ptr(nc), _ptr(ni),
                                                       B, N, M, K, Cc, int(h), int(w), _stream()), "window_match_fwd")
    return dict(conf_matrix=conf, next_conf=nc, next_idx=ni)


def nms_select(next_conf01, next_idx01, next_idx10, hw0, hw1, nms_window=5, test_thr=0.2, pre=(), border_rm=0,
               valid_hw=None, double_check=True, extra_keep=None):
    """pre: sequence of (pre_conf [B,hp*wp], (hp,wp), pre_thr), at most 2.  Returns dict with device count `n`."""
    _chk(next_conf01, "next_conf01"), _chk(next_idx01, "next_idx01", torch.int64), _chk(next_idx10, "next_idx10", torch.int64)
    _chk(valid_hw, "valid_hw", torch.int32)
    extra_keep = _u8(extra_keep)
    B, N = next_conf01.shape
    dev = next_conf01.device
    pre = list(pre)
    if len(pre) > 2:
        raise RuntimeError("at most two previous stages are supported")
    for p in pre:
        _chk(p[0], "pre_conf")
    pre += [(None, (1, 1), 0.0)] * (2 - len(pre))
    l = _lib.lib()
    ws = torch.empty(l.casmtr_nms_select_ws_bytes(B, hw0[0], hw0[1]), device=dev, dtype=torch.uint8)
    bi, ii, ji = (torch.empty(B * N, device=dev, dtype=torch.int64) for _ in range(3))
    mc = torch.empty(B * N, device=dev, dtype=torch.float32)
    n = torch.zeros(1, device=dev, dtype=torch.int64)
    with torch.cuda.device(dev):
        _lib.check(l.casmtr_nms_select_fwd(_ptr(next_conf01), _ptr(next_idx01), _ptr(next_idx10), int(nms_window),
                                           float(test_thr), _ptr(pre[0][0]), pre[0][1][0], pre[0][1][1], float(pre[0][2]),
                                           _ptr(pre[1][0]), pre[1][1][0], pre[1][1][1], float(pre[1][2]), int(border_rm),
                                           _ptr(valid_hw), int(bool(double_check)), _ptr(ws), _ptr(bi), _ptr(ii), _ptr(ji),
                                           _ptr(mc), _ptr(n), B, hw0[0], hw0[1], hw1[0], hw1[1], _ptr(extra_keep),
                                           _stream()),
                   "nms_select_fwd")
    return dict(b_ids=bi, i_ids=ii, j_ids=ji, mconf=mc, n=n, keep_ws=ws)
