"""autograd.Function surface of the QuadTreeAttention ops.

Mirrors cuda_imp/QuadTreeAttention/QuadtreeAttention/functions/quadtree_attention.py:7-57 (same class names, same
argument order, same saved tensors) over the HIP primitives; `score_computation_op` / `value_aggregation_op` are the
names the reference's modules import (modules/quadtree_attention.py:5).
"""
import torch
from torch.autograd import Function

from .. import ops


class ScoreComputation(Function):
    """query [B,N1,4,H,D], key [B,N2,H,D], index [B,N1,K,H] -> [B,N1,4,K,H]"""

    @staticmethod
    def forward(ctx, query, key, index):
        out = ops.qta_score_fwd(query, key, index)
        ctx.save_for_backward(query, key, index)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        query, key, index = ctx.saved_tensors
        dq, dk = ops.qta_score_bwd(grad_output.contiguous(), query, key, index)
        return dq, dk, None


score_computation_op = ScoreComputation.apply


class value_aggregation(Function):
    """score/index [B,N,f,K,H], value [B,M,H,D] -> [B,N,f,H,D]  (the (n f) flattening of :31-38 is a free view)"""

    @staticmethod
    def forward(ctx, score, value, index):
        ctx.save_for_backward(score, value, index)
        B, N, f, K, H = score.shape
        D = value.shape[-1]
        out = torch.empty((B, N * f, H, D), device=score.device, dtype=score.dtype)
        ops.qta_value_agg_fwd(score.reshape(B, N * f, K, H), value, index.reshape(B, N * f, K, H).contiguous(), out)
        return out.view(B, N, f, H, D)

    @staticmethod
    def backward(ctx, grad_output):
        score, value, index = ctx.saved_tensors
        B, N, f, K, H = score.shape
        grad_score = torch.zeros((B, N * f, K, H), device=score.device, dtype=score.dtype)
        grad_value = torch.zeros_like(value)
        ops.qta_value_agg_bwd(grad_output.contiguous().view(B, N * f, H, -1), score.reshape(B, N * f, K, H), value,
                              index.reshape(B, N * f, K, H).contiguous(), grad_score, grad_value)
        return grad_score.view(B, N, f, K, H), grad_value, None


value_aggregation_op = value_aggregation.apply
