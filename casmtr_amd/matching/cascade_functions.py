"""Counterpart of src/model/functions/cascade_functions.py for the pieces on the hot path:
`ScoreComputation` (:8-22) over the HIP window-score kernels, and the border-mask helpers (:82-172)."""
import torch
from torch.autograd import Function

from .. import ops


class ScoreComputation(Function):
    """query [B,N1,C], key [B,N2,C], index [B,N1,K] -> [B,N1,K]"""

    @staticmethod
    def forward(ctx, query, key, index):
        assert query.shape[1] % 16 == 0  # same preconditions as the reference (:11-12)
        assert query.shape[1] / 16 <= 32768
        out = ops.window_score_fwd(query, key, index)
        ctx.save_for_backward(query, key, index)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        query, key, index = ctx.saved_tensors
        dq, dk = ops.window_score_bwd(grad_output.contiguous(), query, key, index)
        return dq, dk, None


def valid_extents(p_m0, p_m1):
    """[B,H,W] padding masks -> [B,4] int32 (h0,w0,h1,w1), the per-pair valid extents used by the *_with_padding
    helpers (:108-109,155-156)."""
    h0, w0 = p_m0.sum(1).max(-1)[0], p_m0.sum(-1).max(-1)[0]
    h1, w1 = p_m1.sum(1).max(-1)[0], p_m1.sum(-1).max(-1)[0]
    return torch.stack([h0, w0, h1, w1], dim=1).to(torch.int32).contiguous()


def mask_border(m, b: int, v):
    """m [N,H0,W0,H1,W1], in place (:82-100)."""
    if b <= 0:
        return
    for d in range(1, 5):
        sl = [slice(None)] * 5
        sl[d] = slice(None, b)
        m[tuple(sl)] = v
        sl[d] = slice(-b, None)
        m[tuple(sl)] = v


def mask_border_with_padding(m, bd, v, p_m0, p_m1):
    """(:103-117)"""
    if bd <= 0:
        return
    for d in range(1, 5):
        sl = [slice(None)] * 5
        sl[d] = slice(None, bd)
        m[tuple(sl)] = v
    ext = valid_extents(p_m0, p_m1).tolist()
    for b_idx, (h0, w0, h1, w1) in enumerate(ext):
        m[b_idx, h0 - bd:] = v
        m[b_idx, :, w0 - bd:] = v
        m[b_idx, :, :, h1 - bd:] = v
        m[b_idx, :, :, :, w1 - bd:] = v


def mask_window_border(mask, idx_2d, b, v, H1, W1):
    """mask [B,H0,W0], idx_2d [B,H0,W0,2] (y,x) (:120-142)"""
    if b <= 0:
        return mask
    mask[:, :b] = v
    mask[:, :, :b] = v
    mask[:, -b:] = v
    mask[:, :, -b:] = v
    y, x = idx_2d[..., 0], idx_2d[..., 1]
    mask[(x < b) | (x > W1 - b) | (y < b) | (y > H1 - b)] = v
    return mask


def mask_window_border_with_padding(mask, idx_2d, b, v, p_m0, p_m1):
    """(:145-172)"""
    if b <= 0:
        return mask
    mask[:, :b] = v
    mask[:, :, :b] = v
    ext = valid_extents(p_m0, p_m1).tolist()
    for b_idx, (h0, w0, h1, w1) in enumerate(ext):
        mask[b_idx, h0 - b:] = v
        mask[b_idx, :, w0 - b:] = v
        y, x = idx_2d[b_idx, ..., 0], idx_2d[b_idx, ..., 1]
        mask[b_idx, (x < b) | (x > w1 - b) | (y < b) | (y > h1 - b)] = v
    return mask
