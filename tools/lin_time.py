"""Where the split projections' time goes (linear16s_kernel, csrc/callers.hip): the four launch shapes of the callers leg, isolated,
with parts of the kernel switched off (CASMTR_LIN_FLAGS: 1 no stores, 2 no MFMAs beyond the first stage pair, 4 rows not loaded),
next to a device copy of the same bytes.  usage: python tools/lin_time.py"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from casmtr_amd import ops

g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")


def timed(run, n=20):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cases = []
# QTAttB layer on the paired batch (16 images of 104 x 104 tokens, C = 256)
B, side, C = 16, 104, 256
x = rn(B, side * side, C)
ws = [rn(C, C) * 0.05 for _ in range(3)]
pre = [ops.prepare_split_weight(w) for w in ws]
cases.append(("qta q,k,v (one x, 3 problems, quads)", lambda: ops.linear_quads_multi([x, x, x], ws, None, side, side, gemm="split", preps=pre),
              x.numel() * 4 * 4))
cases.append(("qta merge (1 problem, tokens)", lambda: ops.linear_multi([x], ws[:1], None, gemm="split", preps=pre[:1]), x.numel() * 4 * 2))
# cascade layer, one direction (8 images of 208 x 208 tokens, C = 128)
B2, side2, C2 = 8, 208, 128
x2, t2 = rn(B2, side2 * side2, C2), rn(B2, side2 * side2, C2)
ws2 = [rn(C2, C2) * 0.05 for _ in range(3)]
pre2 = [ops.prepare_split_weight(w) for w in ws2]
cases.append(("cascade q | k,v (2 x, 3 problems, quads)",
              lambda: ops.linear_quads_multi([x2, t2, t2], ws2, None, side2, side2, gemm="split", preps=pre2), x2.numel() * 4 * 5))
cases.append(("cascade merge (1 problem, tokens)", lambda: ops.linear_multi([x2], ws2[:1], None, gemm="split", preps=pre2[:1]), x2.numel() * 4 * 2))

for name, run, nbytes in cases:
    a = torch.empty(nbytes // 8, device="cuda", dtype=torch.float32)
    b = torch.empty_like(a)
    cp = timed(lambda: b.copy_(a))
    out = [f"{name}: {nbytes / 1e6:.0f} MB moved; copy of as many bytes {cp:.1f} us"]
    for fl in ("0", "0", "1", "4", "5"):   # (the first column still carries the allocator's first-use cost of the output shapes)
        os.environ["CASMTR_LIN_FLAGS"] = fl
        out.append(f"persistent kernel, flags {fl}: {timed(run):.1f} us")
    os.environ["CASMTR_LINEAR16"] = "stationary"
    os.environ["CASMTR_LIN_FLAGS"] = "0"
    out.append(f"stationary: {timed(run):.1f} us")
    os.environ["CASMTR_LINEAR16"] = "tile"
    out.append(f"tile kernel: {timed(run):.1f} us")
    del os.environ["CASMTR_LINEAR16"]
    print("; ".join(out), flush=True)
print(f"q,k,v + pyramid (3 levels) in one launch: {timed(lambda: ops.linear_quads_pyramid_multi([x, x, x], ws, None, side, side, 3, preps=pre)):.1f} us", flush=True)
yq = ops.linear_quads_multi([x, x, x], ws, None, side, side, gemm="split", preps=pre)
print(f"quad_pool level 0 -> 1 (3 tensors): {timed(lambda: ops.quad_pool_multi(yq, side, side)):.1f} us", flush=True)
y1 = ops.quad_pool_multi(yq, side, side)
print(f"quad_pool level 1 -> 2 tokens (3 tensors): {timed(lambda: ops.quad_pool_multi(y1, side // 2, side // 2, to_tokens=True)):.1f} us", flush=True)
