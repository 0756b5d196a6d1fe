"""GPU: SURVEY.md §8 f.1 -- the callers of the attention kernels (QuadtreeAttention / CascadeQuadtreeAttention) and the two
kernels they add (token-major projection GEMM, token pyramid pooling): bit-exact against the oracle, within fp32
tolerance of the reference-python fixtures."""
import numpy as np
import pytest
import torch

import oracle
from golden_inputs import CASES, make_inputs
from parity_utils import assert_close, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("M,Nn,K,nprob,bias", [(300, 256, 256, 3, False), (130, 64, 64, 1, True), (1024, 128, 128, 2, True),
                                               (257, 200, 96, 4, True)])
def test_linear_bit_exact(M, Nn, K, nprob, bias):
    from casmtr_amd import ops
    r = np.random.default_rng(M + Nn)
    xs = [r.standard_normal((M, K)).astype(np.float32) for _ in range(nprob)]
    ws = [r.standard_normal((Nn, K)).astype(np.float32) for _ in range(nprob)]
    bs = [r.standard_normal(Nn).astype(np.float32) if bias and i != 1 else None for i in range(nprob)]
    ys = ops.linear_multi([T(x) for x in xs], [T(w) for w in ws], [None if b is None else T(b) for b in bs])
    for x, w, b, y in zip(xs, ws, bs, ys):
        assert np.array_equal(N(y), oracle.linear(x, w, b)), "projection GEMM differs from the k-ordered fmaf chain"


def test_linear_contract_errors():
    from casmtr_amd import ops
    x, w = torch.zeros(4, 48, device=DEV), torch.zeros(8, 48, device=DEV)
    with pytest.raises(RuntimeError, match="UNSUPPORTED"):
        ops.linear(x, w)                      # K % 32 != 0
    with pytest.raises(RuntimeError):
        ops.linear(x.cpu(), w)                # host tensor: no CPU fallback


@pytest.mark.parametrize("B,h,w,C,n", [(2, 12, 10, 16, 3), (1, 7, 9, 8, 1), (2, 26, 26, 256, 3)])
def test_token_pool_bit_exact(B, h, w, C, n):
    from casmtr_amd import ops
    r = np.random.default_rng(h * w)
    xs = [r.standard_normal((B, h * w, C)).astype(np.float32) for _ in range(n)]
    ys = ops.token_pool_multi([T(x) for x in xs], h, w)
    for x, y in zip(xs, ys):
        assert np.array_equal(N(y), oracle.token_pool(x, h, w))


def _load_weights(m, inp, C, bias):
    with torch.no_grad():
        for n, conv in (("q", m.q_proj), ("k", m.k_proj), ("v", m.v_proj)):
            conv.weight.copy_(T(inp["w" + n]).view(C, C, 1, 1))
            if bias:
                conv.bias.copy_(T(inp["b" + n]))
        m.proj.weight.copy_(T(inp["wp"]))
        m.proj.bias.copy_(T(inp["bp"]))


@pytest.mark.parametrize("name", [n for n, c in CASES["quadtree_block"].items() if c["kind"] == "qta"])
def test_quadtree_attention_block(name):
    from casmtr_amd.modules.quadtree_block import QuadtreeAttention
    cfg = CASES["quadtree_block"][name]
    inp, g = make_inputs("quadtree_block", name), load_golden("quadtree_block", name)
    C, bias = cfg["nhead"] * cfg["D"], bool(cfg.get("qkv_bias"))
    (h, w), (h1, w1) = cfg["hw"], cfg.get("hw1", cfg["hw"])
    m = QuadtreeAttention(C, cfg["nhead"], cfg["topks"], qkv_bias=bias, scale=3, attn_type="B").to(DEV).eval()
    _load_weights(m, inp, C, bias)
    with torch.no_grad():
        m.py_att.weight.copy_(T(inp["weight"]))
        out = m(T(inp["x"]), T(inp["target"]), h, w, h1, w1)
    ref, _ = oracle.quadtree_attention_block(inp["x"], inp["target"], (h, w), (h1, w1), inp["wq"], inp["wk"], inp["wv"],
                                             inp["weight"], inp["wp"], inp["bp"], cfg["nhead"], cfg["topks"], 3,
                                             inp.get("bq"), inp.get("bk"), inp.get("bv"))
    # projections, pooling and the level selection are bit-exact; the messages carry __expf-level differences
    assert_close(N(out), ref, 2e-5, "fused block vs oracle")
    assert_close(N(out), g["out"], TOL, "fused block vs reference python")
    # training / autograd: the reference's structure on torch ops + the composed attention
    x = T(inp["x"]).requires_grad_(True)
    out2 = m(x, T(inp["target"]), h, w, h1, w1)
    assert_close(N(out2), g["out"], TOL, "reference-structure path vs reference python")
    out2.sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and m.q_proj.weight.grad is not None


def test_cascade_quadtree_attention_block():
    from casmtr_amd.modules.quadtree_block import CascadeQuadtreeAttention
    name = "cqa_c8_exact"
    cfg = CASES["quadtree_block"][name]
    inp, g = make_inputs("quadtree_block", name), load_golden("quadtree_block", name)
    C = cfg["nhead"] * cfg["D"]
    hc, wc = cfg["coarse_hw"]
    m = CascadeQuadtreeAttention(C, cfg["nhead"]).to(DEV).eval()
    _load_weights(m, inp, C, False)
    tp = T(g["topk_pos"].astype(np.int64))
    with torch.no_grad():
        out, up = m(T(inp["x"]), T(inp["target"]), 2 * hc, 2 * wc, idx=tp)
    assert np.array_equal(N(up), g["upsampled_idx"].astype(np.int64))
    assert_close(N(out), g["out"], TOL, "fused cascade block vs reference python")
    ref, up_o = oracle.cascade_quadtree_attention_block(inp["x"], inp["target"], (2 * hc, 2 * wc), (2 * hc, 2 * wc),
                                                        g["topk_pos"].astype(np.int64), inp["wq"], inp["wk"], inp["wv"],
                                                        inp["wp"], inp["bp"], cfg["nhead"])
    assert np.array_equal(N(up), up_o)
    assert_close(N(out), ref, 2e-5, "fused cascade block vs oracle")
    x = T(inp["x"]).requires_grad_(True)
    out2, up2 = m(x, T(inp["target"]), 2 * hc, 2 * wc, idx=tp)
    assert np.array_equal(N(up2), g["upsampled_idx"].astype(np.int64))
    assert_close(N(out2), g["out"], TOL, "reference-structure path vs reference python")
    out2.sum().backward()
    assert torch.isfinite(x.grad).all()


def test_block_full_size_levels_bit_exact():
    """BASELINE size (104x104 tokens, C=256, topks 32/16/8): q/k/v projections + pyramid + three attention levels on the
    GPU select exactly the oracle's neighbours; the block output agrees to fp32 round-off."""
    from casmtr_amd import ops
    from casmtr_amd.modules.quadtree_block import QuadtreeAttention
    r = np.random.default_rng(2024)
    B, h, w, C, H = 1, 104, 104, 256, 8
    x = r.standard_normal((B, h * w, C)).astype(np.float32)
    tgt = r.standard_normal((B, h * w, C)).astype(np.float32)
    m = QuadtreeAttention(C, H, [32, 16, 8], scale=3).to(DEV).eval()
    with torch.no_grad():   # unit-variance q/k so that the softmaxes are neither flat nor one-hot
        for conv in (m.q_proj, m.k_proj, m.v_proj):
            conv.weight.copy_(torch.randn(C, C, 1, 1, generator=torch.Generator().manual_seed(1)) / C ** 0.5)
        m.proj.weight.copy_(torch.randn(C, C, generator=torch.Generator().manual_seed(2)) / C ** 0.5)
        out = m(T(x), T(tgt), h, w)
        # level-by-level index parity through the ops the block runs
        q, k, v = ops.linear_multi([T(x), T(tgt), T(tgt)], [m.q_proj.weight, m.k_proj.weight, m.v_proj.weight])
    wq, wk, wv = (N(c.weight).reshape(C, C) for c in (m.q_proj, m.k_proj, m.v_proj))
    assert np.array_equal(N(q), oracle.linear(x, wq)) and np.array_equal(N(v), oracle.linear(tgt, wv))
    ref, levels = oracle.quadtree_attention_block(x, tgt, (h, w), (h, w), wq, wk, wv, N(m.py_att.weight), N(m.proj.weight),
                                                  N(m.proj.bias), H, [32, 16, 8], 3)
    with torch.no_grad():
        q1, k1, v1 = ops.token_pool_multi([q, k, v], h, w)
        q2, k2, v2 = ops.token_pool_multi([q1, k1, v1], h // 2, w // 2)
        l0 = ops.qta_coarse_level(q2, k2, v2, H, 32)
        l1 = ops.qta_fine_level(q1, k1, v1, l0["topk_idx"], (52, 52), (52, 52), H, 16)
    assert np.array_equal(N(l0["topk_idx"]), levels[0]["topk_idx"])
    assert np.array_equal(N(l1["topk_idx"]), levels[1]["topk_idx"])
    assert_close(N(out), ref, 2e-5, "full-size block vs oracle")


@pytest.mark.parametrize("B,h,w,Nn,K,nprob,bias", [(2, 12, 10, 64, 64, 3, True), (1, 26, 26, 256, 256, 2, False), (3, 8, 6, 32, 96, 1, True),
                                                   (2, 52, 52, 128, 128, 4, True)])
def test_linear_quads_equals_linear_then_layout(B, h, w, Nn, K, nprob, bias):
    """casmtr_linear_quads_fwd (round 5): the projection GEMM writes the quad-major per-head layout itself -- bit for bit the values
    of casmtr_linear_fwd re-laid by casmtr_tokens_to_quads (M not a multiple of the 128-row tile, several problems per launch)."""
    from casmtr_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + h)
    xs = [torch.randn((B, h * w, K), generator=g).to(DEV) for _ in range(nprob)]
    ws = [torch.randn((Nn, K), generator=g).to(DEV) for _ in range(nprob)]
    bs = [torch.randn(Nn, generator=g).to(DEV) if bias and i != 1 else None for i in range(nprob)]
    want = [ops.tokens_to_quads(y, h, w) for y in ops.linear_multi(xs, ws, bs)]
    got = ops.linear_quads_multi(xs, ws, bs, h, w)
    for a, b in zip(got, want):
        assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("B,h,w,C,n", [(2, 12, 8, 64, 3), (1, 52, 52, 256, 3), (3, 4, 4, 32, 1), (2, 10, 6, 96, 2)])
def test_quad_pool_equals_token_pool(B, h, w, C, n):
    """casmtr_quad_pool_fwd: the pyramid step on quad-major tensors = casmtr_token_pool_fwd on the token-major ones, written token-major
    (the coarsest level) or quad-major again (h, w multiples of 4)."""
    from casmtr_amd import ops
    g = torch.Generator(device="cpu").manual_seed(h * 100 + w)
    toks = [torch.randn((B, h * w, C), generator=g).to(DEV) for _ in range(n)]
    quads = [ops.tokens_to_quads(t, h, w) for t in toks]
    pooled = ops.token_pool_multi(toks, h, w)
    for a, b in zip(ops.quad_pool_multi(quads, h, w, to_tokens=True), pooled):
        assert torch.equal(a, b)
    if h % 4 == 0 and w % 4 == 0:
        for a, b in zip(ops.quad_pool_multi(quads, h, w), pooled):
            assert torch.equal(a, ops.tokens_to_quads(b, h // 2, w // 2))
    else:
        with pytest.raises(RuntimeError):
            ops.quad_pool_multi(quads, h, w)


def test_block_routes_agree(monkeypatch):
    """QuadtreeAttention / CascadeQuadtreeAttention on the quad-major route (projections written quad-major, pyramid on quad-major
    levels, quad-major attention kernels: opt-in, `layout = "quads"`) against the token-major route (the modules' default): the two
    kernel families select the same neighbours; the outputs agree to the softmax tolerance.  The test also checks that the route it
    names is the route that ran (with KW = 1 windows both 'routes' of the cascade block ran the token-major kernel)."""
    from casmtr_amd import ops
    from casmtr_amd.modules.quadtree_block import CascadeQuadtreeAttention, QuadtreeAttention, set_caller_layout
    g = torch.Generator(device="cpu").manual_seed(9)
    B, h, w, C, H = 2, 52, 52, 256, 8
    x, tgt = torch.randn((B, h * w, C), generator=g).to(DEV), torch.randn((B, h * w, C), generator=g).to(DEV)
    m = QuadtreeAttention(C, H, [32, 16, 8], qkv_bias=True, scale=3).to(DEV).eval()
    c = CascadeQuadtreeAttention(128, 4, qkv_bias=True).to(DEV).eval()
    assert m.layout is None and c.layout is None, "blocks are born on the default (token-major) route"
    calls = {}
    for name in ("cascade_attn_quad", "cascade_attn", "linear_quads_multi", "qta_fine_level_quad", "qta_fine_level"):
        def spy(*a, _f=getattr(ops, name), _n=name, **k):
            calls[_n] = calls.get(_n, 0) + 1
            return _f(*a, **k)
        monkeypatch.setattr(ops, name, spy)
    # 5 x 5 windows around random coarse matches (transformer.py:416-440), same grids and different grids (H,W != H1,W1), with and
    # without the relative position bias of the indoor config ([B, nhead, H0*W0, 4*25])
    cases = []
    for (hq, wq), (hk, wk), rel in (((26, 26), (26, 26), False), ((26, 26), (26, 26), True), ((20, 26), (30, 22), False), ((20, 26), (30, 22), True)):
        xc = torch.randn((B, 4 * hq * wq, 128), generator=g).to(DEV)
        tc = torch.randn((B, 4 * hk * wk, 128), generator=g).to(DEV)
        tp = ops.window_warp_idx(torch.randint(0, hk * wk, (B, hq * wq), generator=g).to(DEV), hk, wk, 5)
        assert tuple(tp.shape) == (B, hq * wq, 25, 2)
        rp = torch.randn((B, 4, 4 * hq * wq, 100), generator=g).to(DEV) if rel else None
        cases.append((xc, tc, (2 * hq, 2 * wq), (2 * hk, 2 * wk), tp, rp))
    outs = {}
    for route in ("tokens", "quads"):
        set_caller_layout(m, route), set_caller_layout(c, route)
        calls.clear()
        with torch.no_grad():
            outs[route] = [m(x, tgt, h, w)] + [c(xc, tc, *hwq, *hwk, idx=tp, rel_pos=rp, want_idx=False)[0] for xc, tc, hwq, hwk, tp, rp in cases]
        if route == "quads":
            assert calls.get("cascade_attn_quad") == len(cases) and not calls.get("cascade_attn"), calls
            assert calls.get("linear_quads_multi", 0) >= 1 + len(cases) and calls.get("qta_fine_level_quad") == 2 and not calls.get("qta_fine_level"), calls
        else:
            assert calls.get("cascade_attn") == len(cases) and not calls.get("cascade_attn_quad") and not calls.get("linear_quads_multi"), calls
            assert calls.get("qta_fine_level") == 2 and not calls.get("qta_fine_level_quad"), calls
    names = ["QuadtreeAttention"] + [f"CascadeQuadtreeAttention {hwq}->{hwk}{' rel_pos' if rp is not None else ''}" for _, _, hwq, hwk, _, rp in cases]
    for a, b, what in zip(outs["tokens"], outs["quads"], names):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(a.abs().max())), what


# ---------------------------------------------------------------------------------------------------------------------
# round 6: the projections on the f16 matrix pipe (ops.linear_multi(gemm="split"), csrc/callers.hip linear16_kernel)
def _rows(kind, g, M, K):
    x = torch.randn((M, K), generator=g)
    if kind == "scaled":      # rows spanning 12 orders of magnitude, a few huge / tiny channels inside a row
        x = x * torch.exp(torch.empty((M, 1)).uniform_(-14, 14, generator=g))
        x[:, 3] *= 300.0
        x[:, 7] *= 1e-4
    elif kind == "sparse":    # most channels exactly zero, some rows all zero
        x = x * (torch.rand((M, K), generator=g) < 0.1)
        x[::17] = 0.0
    return x.contiguous()


@pytest.mark.parametrize("kind", ["randn", "scaled", "sparse"])
@pytest.mark.parametrize("M,N,K", [(1000, 128, 128), (5408 + 37, 256, 256), (300, 256, 128), (128, 128, 256)])
@pytest.mark.parametrize("kernel", ["persistent", "stationary", "tile"])
def test_linear_split_error_bound(kind, M, N, K, kernel, monkeypatch):
    """|y_split - y_exact_chain| <= 2^-15 |x_m| |w_n| for every output (the budget derived in csrc/callers.hip), measured with a factor to
    spare; against float64 the split is as close as the fp32 chain itself.  Ragged M (rows beyond the last full 128-row tile), with bias."""
    from casmtr_amd import ops
    if kernel != "persistent":   # the first two split kernels (one workgroup per 128 x 128 output tile; activation-stationary);
        monkeypatch.setenv("CASMTR_LINEAR16", kernel)   # default: the persistent producer / consumer kernel (csrc/linear_pc.hip)
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x, w = _rows(kind, g, M, K), _rows("randn" if kind == "sparse" else kind, g, N, K) * 0.05
    b = torch.randn((N,), generator=g)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    ye = ops.linear_multi([xd], [wd], [bd], gemm="exact")[0]
    ys = ops.linear_multi([xd], [wd], [bd], gemm="split")[0]
    assert ys.shape == ye.shape == (M, N)
    scale = (x.double().norm(dim=1)[:, None] * w.double().norm(dim=1)[None, :]).to(DEV)       # |x_m| |w_n|
    err = (ys.double() - ye.double()).abs()
    # (+ an ulp of the result: where |bias| dwarfs |x||w| the two paths may round v + bias differently)
    ratio = float(((err - 2.0 ** -23 * ye.double().abs()).clamp(min=0) / (2.0 ** -15 * scale + 1e-300)).max())
    assert ratio <= 0.25, f"split vs exact chain: {ratio:.3f} of the 2^-15 |x||w| budget"
    ref = x.double().to(DEV) @ w.double().to(DEV).T + b.double().to(DEV)
    e_split, e_exact = (ys.double() - ref).abs(), (ye.double() - ref).abs()
    bound = 2.0 ** -15 * scale + 2.0 ** -22 * ref.abs()
    assert bool((e_split <= bound).all()) and bool((e_exact <= bound).all())
    # on average the split is as accurate as the fp32 chain (sparse rows: the chain is often exact, so allow 2 % of the budget on top)
    assert float(e_split.sum()) <= 2.0 * float(e_exact.sum()) + 0.02 * float(bound.sum())


def test_linear_split_multi_and_quads():
    """three problems in one launch with k and v projecting the same tokens (one exponent pass for both); the quad-major store of the
    split kernel is the token-major result re-laid (bit-equal); no bias / bias mixed"""
    from casmtr_amd import ops
    g = torch.Generator(device="cpu").manual_seed(77)
    B, h, w, C = 2, 20, 28, 256
    x, t = torch.randn((B, h * w, C), generator=g).to(DEV), torch.randn((B, h * w, C), generator=g).to(DEV)
    ws = [(0.06 * torch.randn((C, C), generator=g)).to(DEV) for _ in range(3)]
    bs = [torch.randn((C,), generator=g).to(DEV), None, torch.randn((C,), generator=g).to(DEV)]
    tok = ops.linear_multi([x, t, t], ws, bs, gemm="split")
    one = [ops.linear(a, wt, bb, gemm="split") for a, wt, bb in zip((x, t, t), ws, bs)]
    for a, b_ in zip(tok, one):
        assert torch.equal(a, b_), "a problem's result does not depend on what shares its launch"
    qm = ops.linear_quads_multi([x, t, t], ws, bs, h, w, gemm="split")
    for a, b_ in zip(qm, tok):
        assert torch.equal(a, ops.tokens_to_quads(b_, h, w))
    ex = ops.linear_multi([x, t, t], ws, bs, gemm="exact")
    for a, b_ in zip(tok, ex):
        assert float((a - b_).abs().max()) <= 1e-5 * float(b_.abs().max())
    # shapes the split kernel does not cover run the exact kernel (same call, no error): N not a multiple of 128
    w96 = (0.06 * torch.randn((96, C), generator=g)).to(DEV)
    assert torch.equal(ops.linear(x, w96, None, gemm="split"), ops.linear(x, w96, None, gemm="exact"))
    # a weight's prepared form depends on its values: a block re-prepares when the parameter changes (version counter)
    from casmtr_amd.modules.quadtree_block import QuadtreeAttention, set_caller_layout
    m = set_caller_layout(QuadtreeAttention(C, 8, [8, 4, 2], scale=3).to(DEV).eval(), "tokens", "split")
    with torch.no_grad():
        y0 = m(x, t, h, w)
        m.proj.weight.mul_(2.0)
        m.proj.bias.mul_(2.0)
        assert torch.allclose(m(x, t, h, w), 2.0 * y0, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("M,N,K,share", [(1000, 256, 256, True), (64 * 256 * 3 + 5, 128, 128, False), (300, 256, 128, True),
                                           (777, 384, 256, False), (63, 128, 256, True), (20000, 512, 128, True)])
def test_linear_persistent_equals_stationary(M, N, K, share, monkeypatch):
    """the persistent producer / consumer kernel multiplies the same products in the same order as the two earlier split kernels: bit-equal
    results, token-major; ragged M, one or two activation tensors per launch, tail units (K = 128: 128-row units), more blocks than CUs"""
    from casmtr_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N)
    x, t = _rows("scaled", g, M, K).to(DEV), _rows("randn", g, M, K).to(DEV)
    ws = [(0.05 * torch.randn((N, K), generator=g)).to(DEV) for _ in range(3)]
    bs = [torch.randn((N,), generator=g).to(DEV), None, torch.randn((N,), generator=g).to(DEV)]
    xs = [x, x, x] if share else [x, t, t]
    got = ops.linear_multi(xs, ws, bs, gemm="split")
    monkeypatch.setenv("CASMTR_LINEAR16", "stationary")
    want = ops.linear_multi(xs, ws, bs, gemm="split")
    for a, b_ in zip(got, want):
        assert torch.equal(a, b_)


@pytest.mark.parametrize("B,h,w,N,K", [(2, 104, 104, 256, 256), (3, 60, 80, 256, 256), (1, 20, 12, 128, 128), (2, 208, 208, 128, 128),
                                       (5, 8, 8, 256, 256), (2, 12, 20, 256, 128), (3, 4, 4, 128, 128), (1, 4, 12, 256, 256)])
@pytest.mark.parametrize("levels", [1, 2, 3])
def test_linear_pyramid_equals_linear_then_pool(B, h, w, N, K, levels, monkeypatch):
    """casmtr_linear_split_pyramid_fwd: projections + avg_pool2d pyramid from one launch == the quad-major projection (earlier kernel)
    followed by quad_pool launches, every level bit for bit; grids that are not multiples of the 8 x 8 tile (60 x 80, 20 x 12: partial
    tiles), one image tile only, mixed bias / no bias, q from one tensor and k, v from another"""
    from casmtr_amd import ops
    g = torch.Generator(device="cpu").manual_seed(h * w + N + levels)
    x, t = _rows("scaled", g, B * h * w, K).view(B, h * w, K).to(DEV), torch.randn((B, h * w, K), generator=g).to(DEV)
    ws = [(0.05 * torch.randn((N, K), generator=g)).to(DEV) for _ in range(3)]
    bs = [torch.randn((N,), generator=g).to(DEV), None, torch.randn((N,), generator=g).to(DEV)]
    got = ops.linear_quads_pyramid_multi([x, t, t], ws, bs, h, w, levels)
    assert got is not None and all(len(lv) == levels for lv in got)
    monkeypatch.setenv("CASMTR_LINEAR16", "stationary")
    want = ops.linear_quads_multi([x, t, t], ws, bs, h, w, gemm="split")
    for i in range(levels):
        for a, b_ in zip(got, want):
            assert a[i].shape == b_.shape and torch.equal(a[i], b_), f"level {i}"
        if i + 1 < levels:
            want = ops.quad_pool_multi(want, h >> i, w >> i, to_tokens=(i + 1 == levels - 1))
    monkeypatch.delenv("CASMTR_LINEAR16")
    # the plain quad-major entry runs the same kernel without the pooled outputs
    for a, b_ in zip(ops.linear_quads_multi([x, t, t], ws, bs, h, w, gemm="split"), got):
        assert torch.equal(a, b_[0])
    # shapes outside the kernel's cover (K = 256 with a 128-column remainder): None, the caller pools in separate launches
    w384 = [(0.05 * torch.randn((384, 256), generator=g)).to(DEV)]
    assert ops.linear_quads_pyramid_multi([torch.zeros((1, 64, 256), device=DEV)], w384, None, 8, 8, 2) is None


def test_fused_pyramid_in_the_block(monkeypatch):
    """QuadtreeAttention on the quad route with split projections: pyramid from the projection's epilogue == separate pooling launches"""
    from casmtr_amd import _lib
    from casmtr_amd.modules.quadtree_block import QuadtreeAttention, set_caller_layout
    g = torch.Generator(device="cpu").manual_seed(11)
    B, h, w, C = 2, 40, 56, 256
    x, tgt = torch.randn((B, h * w, C), generator=g).to(DEV), torch.randn((B, h * w, C), generator=g).to(DEV)
    m = set_caller_layout(QuadtreeAttention(C, 8, [16, 8, 8], qkv_bias=True, scale=3).to(DEV).eval(), "quads", "split")
    outs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("CASMTR_FUSED_PYRAMID", fused)
        _lib.prof_enable(True)
        with torch.no_grad():
            outs.append(m(x, tgt, h, w))
        torch.cuda.synchronize()
        times = _lib.prof_read()
        _lib.prof_enable(False)
        assert (times.get("token_pool", (0, 0))[1] == 0) == (fused == "1"), times.get("token_pool")
    assert torch.equal(outs[0], outs[1])


def test_linear_persistent_stress():
    """random shapes / problem counts / sharing patterns / grids through the persistent kernel's LDS-counter hand-offs, bit for bit
    against the stationary kernel and the pooling launches (tools/lin_stress.py)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "lin_stress.py"), "120", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "120 cases, 0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_block_forward_multi_equals_per_call_forwards():
    """QuadtreeAttention.forward_multi (both directions of a layer: six projections + their pyramids from one launch into doubled-batch
    operands, attention and merge on the doubled batch) == one forward per direction, 'self' and 'cross' layers; and the fall-back"""
    from casmtr_amd import _lib
    from casmtr_amd.modules.quadtree_block import QuadtreeAttention, set_caller_layout
    g = torch.Generator(device="cpu").manual_seed(23)
    B, h, w, C = 2, 24, 40, 256
    x0, x1 = torch.randn((B, h * w, C), generator=g).to(DEV), torch.randn((B, h * w, C), generator=g).to(DEV)
    m = set_caller_layout(QuadtreeAttention(C, 8, [16, 8, 8], qkv_bias=True, scale=3).to(DEV).eval(), "quads", "split")
    with torch.no_grad():
        for calls in ([(x0, x0), (x1, x1)], [(x0, x1), (x1, x0)]):
            _lib.prof_enable(True)
            got = m.forward_multi(calls, h, w)
            torch.cuda.synchronize()
            times = _lib.prof_read()
            _lib.prof_enable(False)
            assert times["linear_nt"][1] == 2 and "token_pool" not in times, times   # one projection launch + one merge
            for a, (x, t) in zip(got, calls):
                assert torch.equal(a, m(x, t, h, w))
        set_caller_layout(m, "tokens", "split")   # not the quad route: per-call forwards
        for a, (x, t) in zip(m.forward_multi([(x0, x1), (x1, x0)], h, w), [(x0, x1), (x1, x0)]):
            assert torch.equal(a, m(x, t, h, w))


def test_cascade_block_forward_multi_equals_per_call_forwards():
    from casmtr_amd import ops
    from casmtr_amd.modules.quadtree_block import CascadeQuadtreeAttention, set_caller_layout
    g = torch.Generator(device="cpu").manual_seed(29)
    B, hq, C = 2, 14, 128
    h = w = 2 * hq
    x0, x1 = torch.randn((B, h * w, C), generator=g).to(DEV), torch.randn((B, h * w, C), generator=g).to(DEV)
    tp = [ops.window_warp_idx(torch.randint(0, hq * hq, (B, hq * hq), generator=g).to(DEV), hq, hq, 5) for _ in range(2)]
    c = set_caller_layout(CascadeQuadtreeAttention(C, 4, qkv_bias=True).to(DEV).eval(), "quads", "split")
    with torch.no_grad():
        calls = [(x0, x1, tp[0]), (x1, x0, tp[1])]
        for a, (x, t, ix) in zip(c.forward_multi(calls, h, w), calls):
            assert torch.equal(a, c(x, t, h, w, idx=ix, want_idx=False)[0])


def test_blocks_with_split_projections(monkeypatch):
    """QuadtreeAttention / CascadeQuadtreeAttention with proj_gemm='split' (what pipeline.HotPath(callers) and model.timing opt into)
    against the exact-chain projections on both routes; the split kernels ran (spy)"""
    from casmtr_amd import _lib, ops
    from casmtr_amd.modules.quadtree_block import CascadeQuadtreeAttention, QuadtreeAttention, set_caller_layout
    g = torch.Generator(device="cpu").manual_seed(5)
    B, h, w, C, H = 2, 52, 52, 256, 8
    x, tgt = torch.randn((B, h * w, C), generator=g).to(DEV), torch.randn((B, h * w, C), generator=g).to(DEV)
    m = QuadtreeAttention(C, H, [32, 16, 8], qkv_bias=True, scale=3).to(DEV).eval()
    c = CascadeQuadtreeAttention(128, 4, qkv_bias=True).to(DEV).eval()
    with torch.no_grad():   # unit-gain projections: softmaxes as sharp as a trained model's
        for blk in (m, c):
            for lin in (blk.q_proj, blk.k_proj, blk.v_proj, blk.proj):
                lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) / lin.weight.shape[1] ** 0.5)
    hq = 26
    xc, tc = torch.randn((B, 4 * hq * hq, 128), generator=g).to(DEV), torch.randn((B, 4 * hq * hq, 128), generator=g).to(DEV)
    tp = ops.window_warp_idx(torch.randint(0, hq * hq, (B, hq * hq), generator=g).to(DEV), hq, hq, 5)
    for route in ("tokens", "quads"):
        outs = {}
        for gm in ("exact", "split"):
            set_caller_layout(m, route, gm), set_caller_layout(c, route, gm)
            _lib.prof_enable(True)
            with torch.no_grad():
                outs[gm] = (m(x, tgt, h, w), c(xc, tc, 2 * hq, 2 * hq, idx=tp, want_idx=False)[0])
            torch.cuda.synchronize()
            syms = _lib.prof_symbols()
            _lib.prof_read()
            _lib.prof_enable(False)
            assert ("linear16" in (syms.get("linear_nt") or "")) == (gm == "split"), (route, gm, syms.get("linear_nt"))
        for a, b_, what in zip(outs["exact"], outs["split"], ("QuadtreeAttention", "CascadeQuadtreeAttention")):
            # the projections differ by ~1e-7 relative; a top-k near-tie decided the other way moves single tokens further
            d = (a - b_).abs()
            assert float((d <= 2e-4 * float(a.abs().max())).float().mean()) >= 0.999, (route, what, float(d.max()))
