"""Stress of the persistent projection kernel's LDS-counter hand-offs (csrc/linear_pc.hip): random shapes, problem counts, sharing patterns and
grids, every result compared bit for bit with the stationary kernel (token mode) / with projection + pooling launches (quad mode).
usage: python tools/lin_stress.py [iterations] [seed]"""
import os, sys, random, torch
sys.path.insert(0, "/root/repo")
from casmtr_amd import ops

it_n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
g = torch.Generator(device="cuda").manual_seed(1)
bad = 0
for it in range(it_n):
    K = rnd.choice([128, 256])
    N = rnd.choice([128, 256, 512] if K == 128 else [256, 512])
    np_ = rnd.randint(1, min(8, 2048 // N))
    quads = rnd.random() < 0.5
    if quads:
        B, h, w = rnd.randint(1, 5), 4 * rnd.randint(1, 30), 4 * rnd.randint(1, 30)
        M = B * h * w
    else:
        M = rnd.choice([1, 63, 64, 65, 1000, 16385, rnd.randint(1, 70000)])
    nx = rnd.randint(1, min(3, np_))
    xsrc = [torch.randn((M, K), generator=g, device="cuda") * (10.0 ** rnd.uniform(-3, 3)) for _ in range(nx)]
    xs = [xsrc[rnd.randrange(nx)] for _ in range(np_)]
    ws = [torch.randn((N, K), generator=g, device="cuda") * 0.05 for _ in range(np_)]
    bs = [torch.randn((N,), generator=g, device="cuda") if rnd.random() < 0.5 else None for _ in range(np_)]
    os.environ.pop("CASMTR_LINEAR16", None)
    if quads:
        levels = rnd.choice([1, 2, 3])
        got = ops.linear_quads_pyramid_multi([x.view(B, h * w, K) for x in xs], ws, bs, h, w, levels)
        torch.cuda.synchronize()
        os.environ["CASMTR_LINEAR16"] = "stationary"
        want = []
        for i0 in range(0, np_, 4):   # the stationary kernel takes four problems
            want += ops.linear_quads_multi([x.view(B, h * w, K) for x in xs[i0:i0 + 4]], ws[i0:i0 + 4], bs[i0:i0 + 4], h, w, gemm="split")
        ok = True
        for l in range(levels):
            ok &= all(torch.equal(a[l], b_) for a, b_ in zip(got, want))
            if l + 1 < levels:
                nxt = []
                for i0 in range(0, np_, 4):
                    nxt += ops.quad_pool_multi(want[i0:i0 + 4], h >> l, w >> l, to_tokens=(l + 1 == levels - 1))
                want = nxt
        what = f"quads B={B} {h}x{w} levels={levels}"
    else:
        got = []
        for i0 in range(0, np_, 8):
            got += ops.linear_multi(xs[i0:i0 + 8], ws[i0:i0 + 8], bs[i0:i0 + 8], gemm="split")
        torch.cuda.synchronize()
        os.environ["CASMTR_LINEAR16"] = "stationary"
        want = []
        for i0 in range(0, np_, 4):
            want += ops.linear_multi(xs[i0:i0 + 4], ws[i0:i0 + 4], bs[i0:i0 + 4], gemm="split")
        ok = all(torch.equal(a, b_) for a, b_ in zip(got, want))
        what = f"tokens M={M}"
    if not ok:
        bad += 1
        print(f"MISMATCH it {it}: K={K} N={N} nprob={np_} distinct x={nx} {what}", flush=True)
print(f"{it_n} cases, {bad} mismatches", flush=True)
