"""nn.Module surface of QuadTreeAttention: QTAttB, CascadeQTAttB (+ QTAttGuided, QTAttA for API completeness).

Drop-in for cuda_imp/QuadTreeAttention/QuadtreeAttention/modules/quadtree_attention.py: same constructor arguments,
same forward signatures, same state-dict keys (`weight`, `get_vs.*`), same return values.

Two execution paths, both on the GPU:
  * fused (inference, the hot path): one HIP kernel per pyramid level (ops.qta_coarse_level / qta_fine_level /
    cascade_attn); the [B,L/4,4,K,H] score / softmax / int64-index intermediates of the reference never exist.
  * composed (autograd, `lepe`, QTAttB `rel_pos`): the reference's op-by-op structure over the HIP primitives
    score_computation_op / value_aggregation_op, which carry backward kernels.
"""
import torch
import torch.nn as nn

from .. import ops
from ..functions.quadtree_attention import score_computation_op, value_aggregation_op


def _tokens(x, nhead):
    """[B,C,h,w] -> [B,h*w,nhead,C/nhead]  (what :165-167 does with rearrange + view + contiguous)"""
    B, C, h, w = x.shape
    if torch.is_grad_enabled() and x.requires_grad:  # differentiable layout change for the composed path
        return x.float().permute(0, 2, 3, 1).reshape(B, h * w, nhead, C // nhead).contiguous()
    return ops.nchw_to_tokens(x.contiguous().float()).view(B, h * w, nhead, C // nhead)


def _quad_order(x, h, w):
    """raster [B,h*w,...] -> quad-major [B,(h/2)*(w/2),4,...]  ("b c h t1 w t2 -> b (h w) (t1 t2) c", :188-189)"""
    B = x.shape[0]
    rest = x.shape[2:]
    x = x.view(B, h // 2, 2, w // 2, 2, *rest)
    perm = (0, 1, 3, 2, 4) + tuple(range(5, 5 + len(rest)))
    return x.permute(*perm).reshape(B, (h // 2) * (w // 2), 4, *rest)


def _raster_order(x, h, w):
    """inverse of _quad_order: [B,(h/2)*(w/2),4,...] -> [B,h*w,...]  ("b (h w) (t1 t2) ... -> b (h t1 w t2) ...", :226)"""
    B = x.shape[0]
    rest = x.shape[3:]
    x = x.view(B, h // 2, w // 2, 2, 2, *rest)
    perm = (0, 1, 3, 2, 4) + tuple(range(5, 5 + len(rest)))
    return x.permute(*perm).reshape(B, h * w, *rest)


def _unquad_rows(x, H):
    """einops "b (H W) (t1 t2) ... -> b (H t1 W t2) ..." with an explicit H (W = L/H), exactly as the reference spells its
    merge step.  Equal to _raster_order when H is the quad-grid height."""
    B, Lq = x.shape[:2]
    rest = x.shape[3:]
    x = x.view(B, H, Lq // H, 2, 2, *rest)
    perm = (0, 1, 3, 2, 4) + tuple(range(5, 5 + len(rest)))
    return x.permute(*perm).reshape(B, Lq * 4, *rest)


def _children(topk_pos, w1, dilated=1):
    """(row,col) on the coarser grid [2,B,N,K,H] -> child indices on the finer grid [B,N,K,4,H]  (:193-199)"""
    r, c = topk_pos[0] * 2, topk_pos[1] * 2
    return torch.stack([(r + x) * w1 + c + y for x in (0, dilated) for y in (0, dilated)], dim=3)


def _needs_autograd(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class QTAttB(nn.Module):
    def __init__(self, nhead, dim, scale, topks=[32, 32, 32, 32], use_dropout=False, attention_dropout=0.1, lepe=False):
        super().__init__()
        self.use_dropout = use_dropout
        self.topks = topks
        self.nhead = nhead
        self.dim = dim
        self.lepe = lepe
        if lepe:  # locally enhanced position encoding (:151-158)
            self.get_vs = nn.ModuleList(
                [nn.Conv2d(dim * nhead, dim * nhead, kernel_size=3, stride=1, padding=1, groups=dim * nhead)
                 for _ in range(scale)])
        self.register_parameter("weight", nn.Parameter(torch.randn(scale)))

    # ---- composed path (reference op structure, differentiable) -------------------------------------------------
    def process_coarse_level(self, query, key, value, topk, rel_pos=None):
        q, k, v = (_tokens(t, self.nhead) for t in (query, key, value))
        QK = torch.einsum("nlhd,nshd->nlsh", q, k) * (1.0 / q.shape[-1] ** 0.5)
        if rel_pos is not None:
            QK = QK + rel_pos
        A = torch.softmax(QK, dim=-2)
        topk_score, topk_idx = torch.topk(A, dim=-2, k=topk, largest=True)
        message = torch.einsum("nlsh,nshd->nlhd", A, v).contiguous()
        return A, message, topk_score, topk_idx

    def process_fine_level(self, query, key, value, topk_score, topk_pos, topk_prev, topk, final=False, rel_pos=None):
        bs, c, h0, w0 = query.shape
        _, _, h1, w1 = key.shape
        k, v = _tokens(key, self.nhead), _tokens(value, self.nhead)
        q = _quad_order(_tokens(query, self.nhead), h0, w0).contiguous()          # [B,L/4,4,H,D]
        idx = _children(topk_pos, w1).reshape(bs, -1, topk_prev * 4, self.nhead).contiguous()  # parent-major
        QK = score_computation_op(q, k, idx) * (1.0 / q.shape[-1] ** 0.5)         # [B,L/4,4,4K,H]
        idx = idx.unsqueeze(2).expand(-1, -1, 4, -1, -1)
        if rel_pos is not None:  # [1,nhead,L,L] -> gathered at the candidates (:211-215)
            rp = rel_pos.expand(bs, -1, -1, -1).reshape(bs, self.nhead, h0, w0, h1 * w1).permute(0, 2, 3, 4, 1)
            rp = _quad_order(rp.reshape(bs, h0 * w0, h1 * w1, self.nhead), h0, w0)
            QK = QK + torch.gather(rp, index=idx, dim=3)
        A = torch.softmax(QK, dim=-2)
        topk_score, topk_i = torch.topk(A, dim=-2, k=topk, largest=True)
        message = value_aggregation_op(A.contiguous(), v, idx.contiguous())       # [B,L/4,4,H,D]
        topk_idx = torch.gather(idx, index=topk_i, dim=-2)
        return A, message, _raster_order(topk_score, h0, w0).contiguous(), _raster_order(topk_idx, h0, w0).contiguous()

    def _forward_composed(self, queries, keys, values, rel_pos):
        messages = []
        topk = self.topks[0]
        topk_score = topk_pos = None
        for i, (query, key, value) in enumerate(zip(reversed(queries), reversed(keys), reversed(values))):
            w = key.shape[3]
            rp = None if rel_pos is None else rel_pos[i]
            if i == 0:
                A, message, topk_score, topk_idx = self.process_coarse_level(query, key, value, topk, rel_pos=rp)
            else:
                topk_prev, topk = topk, self.topks[i]
                A, message, topk_score, topk_idx = self.process_fine_level(
                    query, key, value, topk_score, topk_pos, topk_prev, topk, i == len(queries) - 1, rel_pos=rp)
            messages.append(message)
            topk_pos = torch.stack([torch.div(topk_idx, w, rounding_mode="trunc"), topk_idx % w])
        weight = torch.softmax(self.weight, dim=0)
        final = None
        for i, m in enumerate(messages):
            if self.lepe:
                lp = _tokens(self.get_vs[i](values[-(i + 1)]), self.nhead)
                m = m + (lp if i == 0 else _quad_order(lp, *values[-(i + 1)].shape[-2:]))
            if i == 0:
                final = m * weight[i]
            else:
                hq, wq = queries[-(i + 1)].shape[2:]
                final = _raster_order(final.unsqueeze(2) + m * weight[i], hq, wq)
        return final.contiguous()

    # ---- fused path -------------------------------------------------------------------------------------------
    # test / audit hook: forward_tokens and forward_quads also materialise the per-level top-k tensors the reference keeps internal
    # (:219-227) and leave the per-level results in self._last_levels (tests/test_model_harness.py audits chained near-tie flips)
    keep_levels = False

    def _level_weights(self):
        """softmax(self.weight) as python floats (they are kernel arguments).  Reading them back is a host sync, so the
        result is cached until the parameter is modified (tensor version counter / storage change)."""
        w = self.weight
        tag = (w._version, w.data_ptr(), str(w.device))
        cached = getattr(self, "_lw_cache", None)
        if cached is None or cached[0] != tag:
            cached = (tag, torch.softmax(w.detach().float(), dim=0).tolist())
            self._lw_cache = cached
        return cached[1]

    def _forward_fused(self, queries, keys, values):
        n = len(queries)
        hw_q = [tuple(q.shape[2:]) for q in reversed(queries)]
        hw_k = [tuple(k.shape[2:]) for k in reversed(keys)]
        if self._quad_major_ok(hw_q, hw_k):
            return self._fused_levels_quad(list(zip(reversed(queries), reversed(keys), reversed(values))), hw_q, hw_k)
        # one launch converts all 3 levels x (q,k,v) to token-major rows
        flat = [t.float() for lvl in zip(reversed(queries), reversed(keys), reversed(values)) for t in lvl]
        toks = ops.nchw_to_tokens_multi(flat)
        return self._fused_levels([toks[3 * i:3 * i + 3] for i in range(n)], hw_q, hw_k)

    def _quad_major_ok(self, hw_q, hw_k):
        """The finer levels run on quad-major operands (csrc/fine_quad.hip) whenever their shapes allow it; CASMTR_FINE_KERNEL =
        dma | quad keeps the token-major kernels (tests compare the two)."""
        import os
        if os.environ.get("CASMTR_FINE_KERNEL", "qm") != "qm" or len(hw_q) < 2:
            return False
        n = len(hw_q)
        return all(ops.fine_quad_supported(self.nhead, self.dim, hw_q[i], hw_k[i], self.topks[i - 1], self.topks[i] if i < n - 1 else 0)
                   for i in range(1, n))

    def _fused_levels_quad(self, levels, hw_q, hw_k, want_topk=False):
        """levels: [(q,k,v)] of [B,C,h,w] tensors, COARSEST first.  Coarsest level token-major (dense attention), every finer level
        quad-major per head; the top-k lists travel between the levels as compact int32 tables, the reference's int64
        [B,L,topk,H] tensors (:219-227, internal to the module) are only written when asked for (want_topk: tests)."""
        n = len(levels)
        weight = self._level_weights()
        # ONE layout launch per call: the coarsest level's operands become token-major, every finer level's quad-major
        flat = [t.float() for lvl in levels for t in lvl]
        is_tok = [i < 3 for i in range(len(flat))]
        conv_in = [(t, k) for t, k in zip(flat, is_tok) if not ops._is_channels_last(t)]
        side = None
        if len(conv_in) == len(flat) and n > 1 and _overlap_layout():
            # The coarsest level (three small, latency-bound kernels) needs only its own token-major operands; the layout pass of the
            # finer levels (HBM-bound, ~95 % of the call's layout bytes) runs beside it on a second HIP stream and is joined before
            # the first fine level.
            main = torch.cuda.current_stream()
            side = _side_stream(flat[0].device)
            side.wait_stream(main)
            with torch.cuda.stream(side):   # high priority: its small kernels get the slots the layout pass keeps freeing
                coarse = ops.nchw_to_quads_multi(flat[:3], [True] * 3)
                out = ops.qta_coarse_level(*coarse, self.nhead, self.topks[0], w_level=weight[0], want_message=False, want_tab=True, want_topk=want_topk)
            fine = ops.nchw_to_quads_multi(flat[3:], [False] * (len(flat) - 3))
            conv = iter(coarse + fine)
        else:
            conv = iter(ops.nchw_to_quads_multi([t for t, _ in conv_in], [k for _, k in conv_in])) if conv_in else iter(())
        laid = []
        for t, k in zip(flat, is_tok):   # channels_last tensors are token-major as they stand (views); finer levels: one token -> quad pass
            if ops._is_channels_last(t):
                B, C, h, w = t.shape
                tok = t.permute(0, 2, 3, 1).reshape(B, h * w, C)
                laid.append(tok if k else ops.tokens_to_quads(tok, h, w))
            else:
                laid.append(next(conv))
        (q0, k0, v0), quads = laid[:3], laid[3:]
        if side is None:
            out = ops.qta_coarse_level(q0, k0, v0, self.nhead, self.topks[0], w_level=weight[0], want_message=False, want_tab=True, want_topk=want_topk)
        else:
            torch.cuda.current_stream().wait_stream(side)
            for t in list(out.values()) + [q0, k0, v0]:   # allocated on the side stream, consumed (and later freed) on this one
                if torch.is_tensor(t):
                    t.record_stream(torch.cuda.current_stream())
        return self._finer_levels_quad(out, quads, hw_q, hw_k, weight, want_topk)

    def _finer_levels_quad(self, out, quads, hw_q, hw_k, weight, want_topk=False):
        """levels 1 .. n-1 on quad-major operands (quads: q, k, v per level, coarser first), `out` the coarsest level's result"""
        n = len(hw_q)
        acc, tab = out["acc"], out["topk_tab"]
        per_level = [out]
        for i in range(1, n):
            q, k, v = quads[3 * (i - 1):3 * i]
            topk = self.topks[i] if i < n - 1 else 0   # the reference computes top-k at the finest level too and discards it
            out = ops.qta_fine_level_quad(q, k, v, tab, hw_q[i], hw_k[i], self.nhead, topk, w_level=weight[i], acc_in=acc,
                                          want_message=False, want_topk=want_topk)
            acc, tab = out["acc"], out["topk_tab"]
            per_level.append(out)
        self._last_levels = per_level if want_topk else None
        return acc

    def quads_ok(self, hw_q, hw_k):
        """hw_q / hw_k finest first: can forward_quads run these shapes on the quad-major kernels?"""
        return self._quad_major_ok(list(reversed(hw_q)), list(reversed(hw_k)))

    def forward_quads(self, coarsest, finer, hw_q, hw_k):
        """Entry point for producers that write the operand layouts themselves (casmtr_amd.modules.quadtree_block: the projections
        through ops.linear_quads_multi, the pyramid through ops.quad_pool_multi -- no layout pass): coarsest = (q, k, v) token-major
        [N, h*w, C] of the coarsest level, finer = [(q, k, v)] quad-major [N, H, (h/2)*(w/2), 4, 32] per finer level, FINEST first;
        hw_q / hw_k finest first -> message [N, H*W, nhead, dim].  Inference only."""
        if self.lepe or _needs_autograd(self.weight):
            raise RuntimeError("QTAttB.forward_quads is the inference path (no lepe, no autograd): use forward()")
        hq, hk = list(reversed(hw_q)), list(reversed(hw_k))
        weight = self._level_weights()
        keep = self.keep_levels
        out = ops.qta_coarse_level(*coarsest, self.nhead, self.topks[0], w_level=weight[0], want_message=False, want_tab=True, want_topk=keep)
        quads = [t for lvl in reversed(finer) for t in lvl]
        return self._finer_levels_quad(out, quads, hq, hk, weight, want_topk=keep)

    def forward_multi(self, calls, split_fine=False):
        """calls: list of (queries, keys, values) pyramid triples of identical shapes whose results do not depend on each other -- the two
        directions of a transformer layer (transformer.py:295-300: `layer(feat0, feat1), layer(feat1, feat0)` are computed from the same
        inputs; the 'self' layers likewise).  One layout launch writes all of them into doubled-batch operands and every level kernel
        runs once -> list of messages, equal to [self.forward(*c) for c in calls].
        split_fine: only the layout pass and the coarsest level (a dense tile kernel whose short grid gains from the doubled batch)
        share their launches; the finer levels (persistent gather kernels walking an XCD's L2 slice pair by pair) run once per call on
        their half of the operands."""
        n = len(calls[0][0])
        hw_q = [tuple(q.shape[2:]) for q in reversed(calls[0][0])]
        hw_k = [tuple(k.shape[2:]) for k in reversed(calls[0][1])]
        flat = [[t for lvl in zip(reversed(qs), reversed(ks), reversed(vs)) for t in lvl] for qs, ks, vs in calls]
        same = all(tuple(a.shape) == tuple(b.shape) for f in flat[1:] for a, b in zip(flat[0], f))
        if (len(calls) < 2 or self.lepe or not same or not self._quad_major_ok(hw_q, hw_k) or any(ops._is_channels_last(t) for f in flat for t in f)
                or _needs_autograd(self.weight, *[t for f in flat for t in f])):
            return [self.forward(*c) for c in calls]
        B = flat[0][0].shape[0]
        groups = [[f[j].float() for f in flat] for j in range(3 * n)]
        laid = ops.nchw_to_quads_grouped(groups, [j < 3 for j in range(3 * n)])
        if split_fine:
            return self._run_levels_quad(laid[:3], laid[3:], hw_q, hw_k, False, groups=len(calls))
        acc = self._run_levels_quad(laid[:3], laid[3:], hw_q, hw_k, False)
        return [acc[g * B:(g + 1) * B] for g in range(len(calls))]

    def _run_levels_quad(self, coarse, quads, hw_q, hw_k, want_topk, groups=1):
        n = len(hw_q)
        weight = self._level_weights()
        out = ops.qta_coarse_level(*coarse, self.nhead, self.topks[0], w_level=weight[0], want_message=False, want_tab=True, want_topk=want_topk)
        acc, tab = out["acc"], out["topk_tab"]
        if groups > 1:   # forward_multi(split_fine=True): the operands hold `groups` calls back to back; finer levels per call -> list
            B = acc.shape[0] // groups
            sls = [slice(g * B, (g + 1) * B) for g in range(groups)]
            state = [(acc[sl], tab[sl]) for sl in sls]
            for g, i in [(g, i) for g in range(groups) for i in range(1, n)]:   # call-major (level-major measured the same)
                q, k, v = (x[sls[g]] for x in quads[3 * (i - 1):3 * i])
                a, t = state[g]
                o = ops.qta_fine_level_quad(q, k, v, t, hw_q[i], hw_k[i], self.nhead, self.topks[i] if i < n - 1 else 0,
                                            w_level=weight[i], acc_in=a, want_message=False, want_topk=False)
                state[g] = (o["acc"], o["topk_tab"])
            self._last_levels = None
            return [a for a, _ in state]
        per_level = [out]
        for i in range(1, n):
            q, k, v = quads[3 * (i - 1):3 * i]
            topk = self.topks[i] if i < n - 1 else 0   # the reference computes top-k at the finest level too and discards it
            out = ops.qta_fine_level_quad(q, k, v, tab, hw_q[i], hw_k[i], self.nhead, topk, w_level=weight[i], acc_in=acc,
                                          want_message=False, want_topk=want_topk)
            acc, tab = out["acc"], out["topk_tab"]
            per_level.append(out)
        self._last_levels = per_level if want_topk else None
        return acc

    def fused_levels_with_topk(self, levels, hw_q, hw_k):
        """Measurement / test hook: the fused path on `levels` = [(q,k,v)] of [B,C,h,w] tensors, COARSEST first, with the per-level
        top-k tensors the reference keeps internal (:219-227) materialised -> list of per-level dicts (topk_idx, topk_score, acc, ...),
        through whichever kernels forward() would run on these shapes."""
        if self._quad_major_ok(hw_q, hw_k):
            self._fused_levels_quad(levels, hw_q, hw_k, want_topk=True)
            return self._last_levels
        toks = ops.nchw_to_tokens_multi([t.float() for lvl in levels for t in lvl])
        weight = self._level_weights()
        acc = prev_idx = None
        outs = []
        for i in range(len(levels)):
            q, k, v = toks[3 * i:3 * i + 3]
            if i == 0:
                out = ops.qta_coarse_level(q, k, v, self.nhead, self.topks[0], w_level=weight[0], want_message=False)
            else:
                out = ops.qta_fine_level(q, k, v, prev_idx, hw_q[i], hw_k[i], self.nhead, self.topks[i] if i < len(levels) - 1 else 0,
                                         w_level=weight[i], acc_in=acc, want_message=False)
            acc, prev_idx = out["acc"], out["topk_idx"]
            outs.append(out)
        return outs

    def _fused_levels(self, levels, hw_q, hw_k):
        """levels: [(q,k,v)] of token-major [B,L,C] tensors, COARSEST first; hw_q / hw_k the matching grid sizes."""
        n = len(levels)
        weight = self._level_weights()
        acc = prev_idx = None
        per_level = []
        for i, (q, k, v) in enumerate(levels):
            if i == 0:
                out = ops.qta_coarse_level(q, k, v, self.nhead, self.topks[0], w_level=weight[0], want_message=False)
            else:
                # the reference computes top-k at the finest level too and throws it away (:219-227): skipped here
                topk = self.topks[i] if i < n - 1 else 0
                out = ops.qta_fine_level(q, k, v, prev_idx, hw_q[i], hw_k[i], self.nhead, topk, w_level=weight[i],
                                         acc_in=acc, want_message=False)
            acc, prev_idx = out["acc"], out["topk_idx"]
            per_level.append(out)
        self._last_levels = per_level if self.keep_levels else None
        return acc

    def forward_tokens(self, queries, keys, values, hw_q, hw_k):
        """Token-major entry point (no reference counterpart; used by casmtr_amd.modules.quadtree_block): pyramids of
        [N, h_i*w_i, C] tensors, finest first, with their grid sizes -> message [N, H*W, nhead, dim].  Inference only."""
        if self.lepe or _needs_autograd(self.weight, *queries, *keys, *values):
            raise RuntimeError("QTAttB.forward_tokens is the inference path (no lepe, no autograd): use forward()")
        lv = list(zip(reversed(queries), reversed(keys), reversed(values)))
        return self._fused_levels(lv, list(reversed(hw_q)), list(reversed(hw_k)))

    def forward(self, queries, keys, values, q_mask=None, kv_mask=None, rel_pos=None):
        """queries/keys/values: pyramids of [N,C,H,W], finest first -> message [N, H*W, nhead, dim]  (:231-286).
        q_mask / kv_mask are accepted and ignored, as in the reference."""
        if self.lepe or rel_pos is not None or _needs_autograd(self.weight, *queries, *keys, *values):
            return self._forward_composed(queries, keys, values, rel_pos)
        return self._forward_fused(queries, keys, values)


_SIDE_STREAMS = {}


def _side_stream(device):
    """one auxiliary HIP stream per device (layout passes that overlap the coarsest level)"""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device, priority=-1)
    return st


def _overlap_layout():
    import os
    return os.environ.get("CASMTR_QTA_OVERLAP", "0") == "1"   # measured slower (545 vs 575 pairs/s): the layout pass slows 1.6x when it shares the chip


class QTAttGuided(QTAttB):
    """QTAttB whose first level starts from externally supplied top-k positions (modules/quadtree_attention.py:289-389)."""

    def __init__(self, nhead, dim, scale, topks=[32], use_dropout=False):
        super().__init__(nhead, dim, scale, topks=topks, use_dropout=use_dropout)

    def forward(self, queries, keys, values, q_mask=None, kv_mask=None, rel_pos=None, topk_pos=None):
        messages = []
        topk = self.topks[0]
        for i, (query, key, value) in enumerate(zip(reversed(queries), reversed(keys), reversed(values))):
            w = key.shape[3]
            rp = None if rel_pos is None else rel_pos[i]
            topk_prev, topk = topk, self.topks[i]
            _, message, _, topk_idx = self.process_fine_level(query, key, value, None, topk_pos, topk_prev, topk,
                                                              i == len(queries) - 1, rel_pos=rp)
            messages.append(message)
            topk_pos = torch.stack([torch.div(topk_idx, w, rounding_mode="trunc"), topk_idx % w])
        weight = torch.softmax(self.weight, dim=0)
        final = None
        for i, m in enumerate(messages):
            final = m * weight[i] if i == 0 else final.unsqueeze(2) + m * weight[i]
            # literal reference behaviour (:384): H = queries[-i].shape[2]; for i == 0 that is queries[0], the FINEST
            # level, so the first message stays in quad-major order -- kept bug-for-bug (no shipped config uses Guided)
            final = _unquad_rows(final, queries[-i].shape[2])
        return final.contiguous()


class QTAttA(nn.Module):
    """Variant A (modules/quadtree_attention.py:8-140): the top-k tokens of a level are masked out of that level's
    message and re-attended at the next level, where each parent's score is redistributed over its 4 children
    (softmax over the children, times the parent's score).  No learned level weights.  No shipped config selects it
    (`attn_type='B'` everywhere); it runs on the composed path: HIP primitives (score / value
    aggregation, with backward) + torch softmax / top-k on the GPU."""

    def __init__(self, nhead, dim, topks=[32, 32, 32, 32], scale=None, use_dropout=False, attention_dropout=0.1):
        super().__init__()
        self.use_dropout = use_dropout
        self.topks = topks
        self.nhead = nhead
        self.dim = dim

    def forward(self, queries, keys, values, q_mask=None, kv_mask=None):
        nh = self.nhead
        n_levels = len(queries)
        topk = self.topks[0]
        final = topk_score = topk_pos = None
        for i, (query, key, value) in enumerate(zip(reversed(queries), reversed(keys), reversed(values))):
            bs, c, h, w = key.shape
            k_, v_ = _tokens(key, nh), _tokens(value, nh)
            temp = 1.0 / k_.shape[-1] ** 0.5
            if i == 0:   # full attention, top-k masked out of the message (:24-44)
                q_ = _tokens(query, nh)
                A = torch.softmax(torch.einsum("nlhd,nshd->nlsh", q_, k_) * temp, dim=-2)
                topk_score, topk_idx = torch.topk(A, dim=-2, k=topk, largest=True)
                message = torch.einsum("nlsh,nshd->nlhd", A.scatter(-2, topk_idx, 0.0), v_)
            else:        # (:46-97)
                topk_prev, topk = topk, self.topks[i]
                last = i == n_levels - 1
                h0, w0 = query.shape[2:]
                q_ = _quad_order(_tokens(query, nh), h0, w0).contiguous()
                idx = _children(topk_pos, w).reshape(bs, -1, topk_prev * 4, nh).contiguous()
                QK = score_computation_op(q_, k_.contiguous(), idx).view(bs, -1, 4, topk_prev, 4, nh) * temp
                A = torch.softmax(QK, dim=-2) * topk_score.unsqueeze(-2).unsqueeze(2)   # score redistribution
                A = A.reshape(bs, -1, 4, topk_prev * 4, nh)
                idx5 = idx.unsqueeze(2).expand(-1, -1, 4, -1, -1).contiguous()
                topk_score, topk_i = torch.topk(A, dim=-2, k=topk, largest=True)
                Am = A if last else A.scatter(-2, topk_i, 0.0)
                message = value_aggregation_op(Am.contiguous(), v_.contiguous(), idx5)
                if not last:
                    topk_idx = _raster_order(torch.gather(idx5, index=topk_i, dim=-2), h0, w0)
                    topk_score = _raster_order(topk_score, h0, w0)
            if i == 0:
                final = message
            else:
                final = _raster_order(final.unsqueeze(2) + message, *query.shape[2:])
            if i < n_levels - 1:
                topk_pos = torch.stack([torch.div(topk_idx, w, rounding_mode="trunc"), topk_idx % w])
        return final


class CascadeQTAttB(nn.Module):
    def __init__(self, nhead, dim, dilated, use_dropout=False):
        super().__init__()
        self.use_dropout = use_dropout
        self.nhead = nhead
        self.dim = dim
        self.dilated = 1 if dilated is None else dilated

    def _forward_composed(self, query, key, value, topk_pos, rel_pos):
        bs, c, h0, w0 = query.shape
        _, _, h1, w1 = key.shape
        kw = topk_pos.shape[2]
        k, v = _tokens(key, self.nhead), _tokens(value, self.nhead)
        q = _quad_order(_tokens(query, self.nhead), h0, w0).contiguous()
        pos = topk_pos.permute(3, 0, 1, 2).unsqueeze(-1).expand(-1, -1, -1, -1, self.nhead)  # [2,B,L/4,KW,H]
        idx = torch.clamp(_children(pos, w1, self.dilated), min=0, max=h1 * w1 - 1)
        idx = idx.reshape(bs, -1, kw * 4, self.nhead).contiguous()
        QK = score_computation_op(q, k, idx) * (1.0 / q.shape[-1] ** 0.5)
        if rel_pos is not None:
            rp = rel_pos.view(bs, self.nhead, h0 * w0, kw * 4).permute(0, 2, 3, 1)
            QK = QK + _quad_order(rp, h0, w0)
        A = torch.softmax(QK, dim=-2)
        idx5 = idx.unsqueeze(2).expand(-1, -1, 4, -1, -1).contiguous()
        message = value_aggregation_op(A.contiguous(), v, idx5)
        message = _raster_order(message, h0, w0).reshape(bs, h0 * w0, c)
        return message, _raster_order(idx5[..., 0], h0, w0)

    def forward(self, query, key, value, topk_pos, rel_pos, want_idx=True):
        """query/key/value [N,C,H,W]; topk_pos [N,(H/2)(W/2),KW,2] (row,col) -> message [N,HW,C], upsampled_idx
        [N,HW,4*KW]  (modules/quadtree_attention.py:400-452).
        want_idx=False (no reference counterpart): the second return value is None.  Every cross layer of a
        CascadeFeatureTransformer returns the SAME index tensor (it depends on topk_pos only, transformer.py:549) and only the
        last one is used, so callers that hold topk_pos (ops.WindowIndex) can skip the 8*N*4KW-byte write entirely."""
        if _needs_autograd(query, key, value, rel_pos):
            return self._forward_composed(query, key, value, topk_pos, rel_pos)
        hw_q, hw_k = tuple(query.shape[2:]), tuple(key.shape[2:])
        import os
        if (not want_idx and os.environ.get("CASMTR_CASCADE_KERNEL", "qm") == "qm"
                and ops.cascade_quad_supported(self.nhead, self.dim, hw_q, hw_k, topk_pos.shape[2], self.dilated)):
            # default inference path: quad-major operands, pairs of query quads sharing one gathered window box (csrc/cascade_quad.hip)
            ts = [t.float() for t in (query, key, value)]
            if all(ops._is_channels_last(t) for t in ts):
                qm = [ops.tokens_to_quads(t.permute(0, 2, 3, 1).reshape(t.shape[0], -1, t.shape[1]), *t.shape[2:]) for t in ts]
            else:
                qm = ops.nchw_to_quads_multi([t.contiguous() for t in ts])
            rp = None if rel_pos is None else rel_pos.contiguous().float()
            return ops.cascade_attn_quad(qm[0], qm[1], qm[2], topk_pos.contiguous(), hw_q, hw_k, self.nhead, rp), None
        q, k, v = ops.nchw_to_tokens_multi([t.float() for t in (query, key, value)])
        return self.forward_tokens(q, k, v, hw_q, hw_k, topk_pos, rel_pos, want_idx)

    def forward_multi(self, calls, split_attn=False):
        """calls: list of (query, key, value, topk_pos) of identical shapes, independent of each other (the two directions of a cascade
        cross layer, transformer.py:549), no rel_pos, no index output: one layout launch, one attention launch on the doubled batch
        (split_attn: one attention launch per call) -> list of messages."""
        q0, k0 = calls[0][0], calls[0][1]
        hw_q, hw_k = tuple(q0.shape[2:]), tuple(k0.shape[2:])
        import os
        ok = (len(calls) > 1 and os.environ.get("CASMTR_CASCADE_KERNEL", "qm") == "qm"
              and ops.cascade_quad_supported(self.nhead, self.dim, hw_q, hw_k, calls[0][3].shape[2], self.dilated)
              and all(tuple(c[j].shape) == tuple(calls[0][j].shape) for c in calls for j in range(4))
              and not any(ops._is_channels_last(t) for c in calls for t in c[:3]) and not _needs_autograd(*[t for c in calls for t in c[:3]]))
        if not ok:
            return [self.forward(q, k, v, tp, None, want_idx=False)[0] for q, k, v, tp in calls]
        B = q0.shape[0]
        qm = ops.nchw_to_quads_grouped([[c[j].float() for c in calls] for j in range(3)])
        if split_attn:   # one layout launch for both directions, the attention per direction on its half of the operands
            return [ops.cascade_attn_quad(qm[0][g * B:(g + 1) * B], qm[1][g * B:(g + 1) * B], qm[2][g * B:(g + 1) * B], c[3].contiguous(),
                                          hw_q, hw_k, self.nhead, None) for g, c in enumerate(calls)]
        tp = torch.cat([c[3].contiguous() for c in calls], 0)
        msg = ops.cascade_attn_quad(qm[0], qm[1], qm[2], tp, hw_q, hw_k, self.nhead, None)
        return [msg[g * B:(g + 1) * B] for g in range(len(calls))]

    def quads_ok(self, hw_q, hw_k, kw):
        import os
        return (os.environ.get("CASMTR_CASCADE_KERNEL", "qm") == "qm"
                and ops.cascade_quad_supported(self.nhead, self.dim, tuple(hw_q), tuple(hw_k), kw, self.dilated))

    def forward_quads(self, q, k, v, hw_q, hw_k, topk_pos, rel_pos=None):
        """Quad-major entry point (operands [N, H, (h/2)*(w/2), 4, 32] written by the projections themselves,
        ops.linear_quads_multi): message only, no index output.  Inference only."""
        rp = None if rel_pos is None else rel_pos.contiguous().float()
        return ops.cascade_attn_quad(q, k, v, topk_pos.contiguous(), tuple(hw_q), tuple(hw_k), self.nhead, rp)

    def forward_tokens(self, q, k, v, hw_q, hw_k, topk_pos, rel_pos=None, want_idx=True):
        """Token-major entry point: q [N,h0*w0,C], k/v [N,h1*w1,C].  Inference only."""
        if _needs_autograd(q, k, v, rel_pos):
            raise RuntimeError("CascadeQTAttB.forward_tokens is the inference path: use forward() for autograd")
        rp = None if rel_pos is None else rel_pos.contiguous().float()
        return ops.cascade_attn(q, k, v, topk_pos.contiguous(), tuple(hw_q), tuple(hw_k), self.nhead, self.dilated, rp,
                                want_idx=want_idx)
