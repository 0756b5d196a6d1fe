"""QuadtreeAttention / CascadeQuadtreeAttention: the modules that CALL QTAttB / CascadeQTAttB.

Drop-in for src/model/modules/quadtree_attention.py: same constructor arguments, same forward signatures, same
state-dict keys (`q_proj.weight`, `k_proj.weight`, `v_proj.weight` [C,C,1,1], `py_att.weight` / `cross_attn.*`,
`proj.weight`, `proj.bias`).

The reference takes [B,N,C] tokens, permutes them to NCHW (+ contiguous), applies three 1x1 convolutions, builds the
avg-pool pyramid in NCHW, and QTAttB turns every level back into tokens.  The inference path here never leaves the token
layout: one batched fp32-MFMA GEMM launch for the three projections, one pooling launch per pyramid level for q/k/v
together, the fused level kernels on those buffers, one GEMM for the output projection.  No layout kernel runs at all.
Two routes do that (see _quad_route): token-major throughout (the default: exact expf / division softmax, what the chained
reference-parity tests pin), or quad-major operands written by the projections themselves (opt-in: the hot path's kernels).

With autograd (training), `attn_type` 'A' / 'Guided', `lepe` or a QTAttB `rel_pos`, forward() keeps the reference's
structure on torch ops + the composed attention modules.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .quadtree_attention import CascadeQTAttB, QTAttA, QTAttB, QTAttGuided, _needs_autograd


def _init_weights(m):
    """src/model/modules/quadtree_attention.py:50-66 (timm's trunc_normal_ == torch.nn.init.trunc_normal_)."""
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)
    elif isinstance(m, nn.Conv2d):
        nn.init.trunc_normal_(m.weight, std=0.02)
        m.init = True  # QuadtreeBlock._init_weights skips modules carrying this flag (transformer.py:176-177)
        if m.bias is not None:
            m.bias.data.zero_()


def _preps(mod, lins):
    """prepared (f16-split) weights of `lins` when the block's projections run on the split path, else None.  Cached on the module per
    layer, re-prepared when the parameter's storage or version counter changes (load_state_dict, an optimiser step, .to(device))."""
    if ops.linear_gemm_mode(mod.proj_gemm) != "split":
        return None
    cache = mod.__dict__.setdefault("_split_prep", {})
    out = []
    for lin in lins:
        w = lin.weight
        tag = (w.data_ptr(), w._version, str(w.device), w.dtype)
        hit = cache.get(id(lin))
        if hit is None or hit[0] != tag:
            hit = (tag, ops.prepare_split_weight(w.detach().float().reshape(w.shape[0], -1).contiguous()) if w.is_cuda else None)
            cache[id(lin)] = hit
        out.append(hit[1])
    return None if any(p is None for p in out) else out


def _project_qkv(mod, x, target):
    """q_proj(x), k_proj(target), v_proj(target) on tokens, one launch."""
    ws = [mod.q_proj.weight, mod.k_proj.weight, mod.v_proj.weight]
    bs = [mod.q_proj.bias, mod.k_proj.bias, mod.v_proj.bias]
    g = mod.proj_gemm
    pp = _preps(mod, [mod.q_proj, mod.k_proj, mod.v_proj])
    if x.shape == target.shape:
        return ops.linear_multi([x, target, target], [w.detach().float() for w in ws],
                                [None if b is None else b.detach().float() for b in bs], gemm=g, preps=pp)
    # different token counts (H,W != H1,W1): the query projection is its own problem shape
    q = ops.linear(x, ws[0].detach().float(), None if bs[0] is None else bs[0].detach().float(), gemm=g, prep=pp[0] if pp else None)
    k, v = ops.linear_multi([target, target], [w.detach().float() for w in ws[1:]],
                            [None if b is None else b.detach().float() for b in bs[1:]], gemm=g, preps=pp[1:] if pp else None)
    return q, k, v


def _project_qkv_quads(mod, x, target, hw, hw1):
    """the same projections written quad-major per head by the GEMM itself (ops.linear_quads_multi): no token -> quad layout pass"""
    ws = [mod.q_proj.weight, mod.k_proj.weight, mod.v_proj.weight]
    bs = [mod.q_proj.bias, mod.k_proj.bias, mod.v_proj.bias]
    ws = [w.detach().float() for w in ws]
    bs = [None if b is None else b.detach().float() for b in bs]
    g = mod.proj_gemm
    pp = _preps(mod, [mod.q_proj, mod.k_proj, mod.v_proj])
    if x.shape == target.shape and tuple(hw) == tuple(hw1):
        return ops.linear_quads_multi([x, target, target], ws, bs, *hw, gemm=g, preps=pp)
    (q,) = ops.linear_quads_multi([x], ws[:1], bs[:1], *hw, gemm=g, preps=pp[:1] if pp else None)
    k, v = ops.linear_quads_multi([target, target], ws[1:], bs[1:], *hw1, gemm=g, preps=pp[1:] if pp else None)
    return q, k, v


def _project_qkv_pyramid(mod, x, target, hw, hw1):
    """projections + avg-pool pyramid in one launch (ops.linear_quads_pyramid_multi) -> ((q, k, v) of the coarsest level, token-major;
    [(q, k, v) per finer level, quad-major, finest first]), or None when that kernel does not serve the module (exact GEMM requested,
    more than three levels, shapes it does not cover)"""
    import os
    if ops.linear_gemm_mode(mod.proj_gemm) != "split" or mod.scale > 3 or os.environ.get("CASMTR_FUSED_PYRAMID", "1") == "0":
        return None
    ws = [w.detach().float() for w in (mod.q_proj.weight, mod.k_proj.weight, mod.v_proj.weight)]
    bs = [None if b is None else b.detach().float() for b in (mod.q_proj.bias, mod.k_proj.bias, mod.v_proj.bias)]
    pp = _preps(mod, [mod.q_proj, mod.k_proj, mod.v_proj])
    if x.shape == target.shape and tuple(hw) == tuple(hw1):
        r = ops.linear_quads_pyramid_multi([x, target, target], ws, bs, *hw, mod.scale, preps=pp)
        if r is None:
            return None
        q, k, v = r
    else:
        rq = ops.linear_quads_pyramid_multi([x], ws[:1], bs[:1], *hw, mod.scale, preps=pp[:1] if pp else None)
        rk = ops.linear_quads_pyramid_multi([target, target], ws[1:], bs[1:], *hw1, mod.scale, preps=pp[1:] if pp else None)
        if rq is None or rk is None:
            return None
        (q,), (k, v) = rq, rk
    return (q[-1], k[-1], v[-1]), [(q[i], k[i], v[i]) for i in range(mod.scale - 1)]


def _quad_route(mod):
    """Which kernels serve a block's inference forward.  "tokens" (default): token-major projections and the token-major attention
    kernels, whose softmax uses expf and a true division -- the route the chained reference-parity tests hold to 3e-3
    (tests/test_model_harness.py).  "quads": the projections write the quad-major operand layout and the quad-major kernels of the
    hot path run (hardware exponential + reciprocal: per call the same indices, values within 2e-5 of the other route; over a chain
    of layers those 1e-6 can decide a top-k near-tie the other way).  The throughput-oriented callers opt in explicitly:
    pipeline.HotPath (cfg.caller_layout), model.timing, set_caller_layout(); CASMTR_CALLER_LAYOUT sets the default for blocks that
    were not told."""
    import os
    return (mod.layout or os.environ.get("CASMTR_CALLER_LAYOUT", "tokens")) == "quads"


def set_caller_layout(module, layout, proj_gemm="keep"):
    """layout 'tokens' | 'quads' | None (= the CASMTR_CALLER_LAYOUT default) for every QuadtreeAttention / CascadeQuadtreeAttention
    inside `module`; proj_gemm 'exact' | 'split' | None (= ops.linear_gemm_mode's default) additionally selects how their four
    projections multiply ('keep': unchanged).  Returns the module."""
    if layout not in ("tokens", "quads", None):
        raise ValueError(f"caller layout {layout!r} (tokens | quads | None)")
    if proj_gemm not in ("exact", "split", None, "keep"):
        raise ValueError(f"projection gemm {proj_gemm!r} (exact | split | None)")
    for m in module.modules():
        if isinstance(m, (QuadtreeAttention, CascadeQuadtreeAttention)):
            m.layout = layout
            if proj_gemm != "keep":
                m.proj_gemm = proj_gemm
    return module


class QuadtreeAttention(nn.Module):
    def __init__(self, dim, num_heads, topks, value_branch=False, act=nn.GELU(), qkv_bias=False, qk_scale=None,
                 attn_drop=0.0, proj_drop=0.0, scale=1, attn_type="B"):
        super().__init__()
        assert dim % num_heads == 0, f"dim {dim} should be divided by num_heads {num_heads}."
        self.dim = dim
        self.num_heads = num_heads
        self.q_proj = nn.Conv2d(dim, dim, kernel_size=1, stride=1, bias=qkv_bias)
        self.k_proj = nn.Conv2d(dim, dim, kernel_size=1, stride=1, bias=qkv_bias)
        self.v_proj = nn.Conv2d(dim, dim, kernel_size=1, stride=1, bias=qkv_bias)
        self.attn_type = attn_type
        if attn_type == "Guided":
            self.py_att = QTAttGuided(num_heads, dim // num_heads, scale=scale, topks=topks)
        elif attn_type == "A":
            self.py_att = QTAttA(num_heads, dim // num_heads, scale=scale, topks=topks)
        else:
            self.py_att = QTAttB(num_heads, dim // num_heads, scale=scale, topks=topks)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.scale = scale
        self.layout = None   # see _quad_route
        self.proj_gemm = None   # "exact" | "split" | None (= ops.linear_gemm_mode default): see set_caller_layout
        self.apply(_init_weights)

    def _fused_ok(self, x, target, rel_pos):
        return (self.attn_type not in ("A", "Guided") and rel_pos is None and x.is_cuda and not self.py_att.lepe
                and not _needs_autograd(x, target, *self.parameters()))

    def forward(self, x, target, H, W, H1=None, W1=None, rel_pos=None, topk_pos=None):
        """x [B,H*W,C], target [B,H1*W1,C] -> [B,H*W,C]  (src/model/modules/quadtree_attention.py:68-100)."""
        H1 = H if H1 is None else H1
        W1 = W if W1 is None else W1
        B, N, C = x.shape
        if not self._fused_ok(x, target, rel_pos):
            return self._forward_reference_structure(x, target, H, W, H1, W1, rel_pos, topk_pos)
        hw_q = [(H >> i, W >> i) for i in range(self.scale)]
        hw_k = [(H1 >> i, W1 >> i) for i in range(self.scale)]
        if (_quad_route(self) and self.scale > 1 and C % 32 == 0 and all(h % 2 == 0 and w % 2 == 0 for h, w in hw_q[:-1] + hw_k[:-1])
                and self.py_att.quads_ok(hw_q, hw_k)):
            # projections straight into the fine-level kernels' quad-major layout, pyramid on quad-major levels, the coarsest level
            # pooled into the token-major layout its kernel reads: no layout pass anywhere (round 5)
            pyr = _project_qkv_pyramid(self, x.contiguous().float(), target.contiguous().float(), (H, W), (H1, W1))
            if pyr is not None:   # split GEMM: the pyramid comes out of the projection's epilogue (one launch)
                (q, k, v), finer = pyr
            else:
                q, k, v = _project_qkv_quads(self, x.contiguous().float(), target.contiguous().float(), (H, W), (H1, W1))
                finer = []
                for i in range(self.scale - 1):
                    finer.append((q, k, v))
                    last = i == self.scale - 2
                    if hw_q[i] == hw_k[i]:
                        q, k, v = ops.quad_pool_multi([q, k, v], *hw_q[i], to_tokens=last)
                    else:
                        (q,), (k, v) = ops.quad_pool_multi([q], *hw_q[i], to_tokens=last), ops.quad_pool_multi([k, v], *hw_k[i], to_tokens=last)
            msg = self.py_att.forward_quads((q, k, v), finer, hw_q, hw_k).view(B, -1, C)
            out = ops.linear(msg, self.proj.weight.detach().float(),
                             None if self.proj.bias is None else self.proj.bias.detach().float(), gemm=self.proj_gemm,
                             prep=(_preps(self, [self.proj]) or [None])[0])
            return self.proj_drop(out)
        q, k, v = _project_qkv(self, x.contiguous().float(), target.contiguous().float())
        queries, keys, values, hw_q, hw_k = [], [], [], [], []
        h, w, h1, w1 = H, W, H1, W1
        for i in range(self.scale):
            queries.append(q), keys.append(k), values.append(v)
            hw_q.append((h, w)), hw_k.append((h1, w1))
            if i != self.scale - 1:
                if (h, w) == (h1, w1):
                    q, k, v = ops.token_pool_multi([q, k, v], h, w)
                else:
                    (q,), (k, v) = ops.token_pool_multi([q], h, w), ops.token_pool_multi([k, v], h1, w1)
                h, w, h1, w1 = h // 2, w // 2, h1 // 2, w1 // 2
        msg = self.py_att.forward_tokens(queries, keys, values, hw_q, hw_k).view(B, -1, C)
        out = ops.linear(msg, self.proj.weight.detach().float(),
                         None if self.proj.bias is None else self.proj.bias.detach().float(), gemm=self.proj_gemm,
                             prep=(_preps(self, [self.proj]) or [None])[0])
        return self.proj_drop(out)

    def forward_multi(self, calls, H, W):
        """calls: [(x, target)] of identical shapes whose results do not depend on each other -- the two directions of a transformer layer
        (transformer.py:295-300) -> [self.forward(x, target, H, W) for x, target in calls], from ONE projection launch (q / k / v of
        every call written into doubled-batch quad-major operands, pyramid included), one attention launch per level and one merge
        projection: the paired launches of QTAttB.forward_multi for callers that enter through the block.  Falls back to per-call
        forwards where the quad route / the fused projection does not apply."""
        n = len(calls)
        x0, t0 = calls[0]
        B, N, C = x0.shape
        hw = [(H >> i, W >> i) for i in range(self.scale)]
        ok = (n > 1 and self.attn_type == "B" and not self.py_att.lepe and self.scale in (2, 3) and x0.is_cuda and _quad_route(self)
              and ops.linear_gemm_mode(self.proj_gemm) == "split" and 3 * n <= 8 and C % 32 == 0
              and all(tuple(x.shape) == (B, N, C) and tuple(t.shape) == (B, N, C) for x, t in calls)
              and all(h % 2 == 0 and w % 2 == 0 for h, w in hw[:-1]) and self.py_att.quads_ok(hw, hw)
              and not _needs_autograd(*[t for c in calls for t in c], *self.parameters()))
        if not ok:
            return [self.forward(x, t, H, W) for x, t in calls]
        lins = [self.q_proj, self.k_proj, self.v_proj]
        ws = [l.weight.detach().float() for l in lins]
        bs = [None if l.bias is None else l.bias.detach().float() for l in lins]
        pp = _preps(self, lins)
        emp = lambda *shape: torch.empty(shape, device=x0.device, dtype=torch.float32)
        Hh = C // 32
        big = []   # per level: (q, k, v) on the doubled batch; finest first; the coarsest token-major
        for l in range(self.scale):
            h, w = hw[l]
            shape = (n * B, h * w, C) if l == self.scale - 1 else (n * B, Hh, (h // 2) * (w // 2), 4, 32)
            big.append([emp(*shape) for _ in range(3)])
        xs, outs = [], []
        for g, (x, t) in enumerate(calls):
            xc, tc = x.contiguous().float(), t.contiguous().float()
            xs += [xc, tc, tc]
            outs += [[big[l][j][g * B:(g + 1) * B] for l in range(self.scale)] for j in range(3)]
        if ops.linear_quads_pyramid_multi(xs, ws * n, bs * n, H, W, self.scale, preps=None if pp is None else list(pp) * n, outs=outs) is None:
            return [self.forward(x, t, H, W) for x, t in calls]
        msg = self.py_att.forward_quads(tuple(big[-1]), [tuple(lv) for lv in big[:-1]], hw, hw).view(n * B, -1, C)
        out = ops.linear(msg, self.proj.weight.detach().float(), None if self.proj.bias is None else self.proj.bias.detach().float(),
                         gemm=self.proj_gemm, prep=(_preps(self, [self.proj]) or [None])[0])
        return [self.proj_drop(out[g * B:(g + 1) * B]) for g in range(n)]

    def _forward_reference_structure(self, x, target, H, W, H1, W1, rel_pos, topk_pos):
        B, N, C = x.shape
        x = x.permute(0, 2, 1).reshape(B, C, H, W).contiguous()
        target = target.permute(0, 2, 1).reshape(B, C, H1, W1).contiguous()
        q, k, v = self.q_proj(x), self.k_proj(target), self.v_proj(target)
        queries, keys, values = [], [], []
        for i in range(self.scale):
            keys.append(k.float()), values.append(v.float()), queries.append(q.float())
            if i != self.scale - 1:
                k, q, v = (F.avg_pool2d(t, kernel_size=2, stride=2) for t in (k, q, v))
        if self.attn_type == "Guided":
            msg = self.py_att(queries, keys, values, rel_pos=rel_pos, topk_pos=topk_pos)
        elif self.attn_type == "A":
            msg = self.py_att(queries, keys, values)
        else:
            msg = self.py_att(queries, keys, values, rel_pos=rel_pos)
        return self.proj_drop(self.proj(msg.reshape(B, -1, C).contiguous()))


class CascadeQuadtreeAttention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0, scale=2, dilated=1):
        super().__init__()
        assert dim % num_heads == 0, f"dim {dim} should be divided by num_heads {num_heads}."
        self.dim = dim
        self.num_heads = num_heads
        self.q_proj = nn.Conv2d(dim, dim, kernel_size=1, stride=1, bias=qkv_bias)
        self.k_proj = nn.Conv2d(dim, dim, kernel_size=1, stride=1, bias=qkv_bias)
        self.v_proj = nn.Conv2d(dim, dim, kernel_size=1, stride=1, bias=qkv_bias)
        self.cross_attn = CascadeQTAttB(num_heads, dim // num_heads, dilated=dilated)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.scale = scale
        self.layout = None   # see _quad_route
        self.proj_gemm = None   # "exact" | "split" | None (= ops.linear_gemm_mode default): see set_caller_layout
        self.apply(_init_weights)

    def forward_multi(self, calls, H, W):
        """calls: [(x, target, idx)] of identical shapes, independent of each other (the two directions of a cascade cross layer,
        transformer.py:549), no rel_pos, no index output -> [self.forward(x, target, H, W, idx=idx, want_idx=False)[0] for ...] from one
        projection launch into doubled-batch quad-major operands, one attention launch and one merge projection."""
        n = len(calls)
        x0, t0, i0 = calls[0]
        B, N, C = x0.shape
        ok = (n > 1 and 3 * n <= 8 and x0.is_cuda and _quad_route(self) and ops.linear_gemm_mode(self.proj_gemm) == "split" and C % 32 == 0
              and H % 2 == 0 and W % 2 == 0 and self.cross_attn.quads_ok((H, W), (H, W), i0.shape[2])
              and all(tuple(x.shape) == (B, N, C) and tuple(t.shape) == (B, N, C) and tuple(ix.shape) == tuple(i0.shape) for x, t, ix in calls)
              and not _needs_autograd(*[t for c in calls for t in c[:2]], *self.parameters()))
        if not ok:
            return [self.forward(x, t, H, W, idx=ix, want_idx=False)[0] for x, t, ix in calls]
        lins = [self.q_proj, self.k_proj, self.v_proj]
        ws = [l.weight.detach().float() for l in lins]
        bs = [None if l.bias is None else l.bias.detach().float() for l in lins]
        pp = _preps(self, lins)
        big = [torch.empty((n * B, C // 32, (H // 2) * (W // 2), 4, 32), device=x0.device, dtype=torch.float32) for _ in range(3)]
        xs, outs = [], []
        for g, (x, t, _) in enumerate(calls):
            xc, tc = x.contiguous().float(), t.contiguous().float()
            xs += [xc, tc, tc]
            outs += [[big[j][g * B:(g + 1) * B]] for j in range(3)]
        if ops.linear_quads_pyramid_multi(xs, ws * n, bs * n, H, W, 1, preps=None if pp is None else list(pp) * n, outs=outs) is None:
            return [self.forward(x, t, H, W, idx=ix, want_idx=False)[0] for x, t, ix in calls]
        tp = torch.cat([ix.contiguous() for _, _, ix in calls], 0)
        msg = self.cross_attn.forward_quads(big[0], big[1], big[2], (H, W), (H, W), tp, None)
        out = ops.linear(msg.view(n * B, -1, C), self.proj.weight.detach().float(),
                         None if self.proj.bias is None else self.proj.bias.detach().float(), gemm=self.proj_gemm,
                         prep=(_preps(self, [self.proj]) or [None])[0])
        return [self.proj_drop(out[g * B:(g + 1) * B]) for g in range(n)]

    def forward(self, x, target, H, W, H1=None, W1=None, idx=None, rel_pos=None, want_idx=True):
        """x [B,H*W,C], target [B,H1*W1,C], idx [B,(H/2)(W/2),KW,2] -> (x' [B,H*W,C], upsampled_idx [B,H*W,4KW])
        (src/model/modules/quadtree_attention.py:152-176).  want_idx=False: see CascadeQTAttB.forward."""
        H1 = H if H1 is None else H1
        W1 = W if W1 is None else W1
        B, N, C = x.shape
        if rel_pos is not None:
            rel_pos = rel_pos.to(torch.float32)
        if x.is_cuda and not _needs_autograd(x, target, rel_pos, *self.parameters()):
            if (not want_idx and _quad_route(self) and C % 32 == 0 and H % 2 == 0 and W % 2 == 0 and H1 % 2 == 0 and W1 % 2 == 0
                    and self.cross_attn.quads_ok((H, W), (H1, W1), idx.shape[2])):
                q, k, v = _project_qkv_quads(self, x.contiguous().float(), target.contiguous().float(), (H, W), (H1, W1))
                msg = self.cross_attn.forward_quads(q, k, v, (H, W), (H1, W1), idx, rel_pos)
                out = ops.linear(msg.view(B, -1, C), self.proj.weight.detach().float(),
                                 None if self.proj.bias is None else self.proj.bias.detach().float(), gemm=self.proj_gemm,
                             prep=(_preps(self, [self.proj]) or [None])[0])
                return self.proj_drop(out), None
            q, k, v = _project_qkv(self, x.contiguous().float(), target.contiguous().float())
            msg, upsampled_idx = self.cross_attn.forward_tokens(q, k, v, (H, W), (H1, W1), idx, rel_pos, want_idx)
            out = ops.linear(msg.view(B, -1, C), self.proj.weight.detach().float(),
                             None if self.proj.bias is None else self.proj.bias.detach().float(), gemm=self.proj_gemm,
                             prep=(_preps(self, [self.proj]) or [None])[0])
            return self.proj_drop(out), upsampled_idx
        x = x.permute(0, 2, 1).reshape(B, C, H, W).contiguous()
        target = target.permute(0, 2, 1).reshape(B, C, H1, W1).contiguous()
        q, k, v = self.q_proj(x), self.k_proj(target), self.v_proj(target)
        msg, upsampled_idx = self.cross_attn(q.float(), k.float(), v.float(), idx, rel_pos)
        return self.proj_drop(self.proj(msg.reshape(B, -1, C).contiguous())), upsampled_idx
