import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from casmtr_amd.modules.quadtree_block import QuadtreeAttention
from casmtr_amd import ops
DEV = "cuda:0"
for (h, w, B, C, H, topks) in [(32, 24, 2, 256, 8, [16, 8, 8]), (32, 24, 1, 256, 8, [16, 8, 8]), (16, 16, 2, 256, 8, [16, 8, 8]), (60, 80, 1, 256, 8, [16, 8, 8]), (52, 52, 2, 256, 8, [32, 16, 8]), (60, 80, 2, 256, 8, [32, 16, 8]), (40, 56, 1, 256, 8, [16, 8, 8])]:
    g = torch.Generator(device="cpu").manual_seed(9)
    x, tgt = torch.randn((B, h * w, C), generator=g).to(DEV), torch.randn((B, h * w, C), generator=g).to(DEV)
    m = QuadtreeAttention(C, H, topks, qkv_bias=True, scale=3).to(DEV).eval()
    outs = {}
    for route in ("tokens", "quads"):
        os.environ["CASMTR_CALLER_LAYOUT"] = route
        with torch.no_grad():
            outs[route] = m(x, tgt, h, w)
    d = (outs["tokens"] - outs["quads"]).abs()
    print((h, w, B, topks), "max diff", float(d.max()), "tokens differing > 1e-4:", int((d.amax(-1) > 1e-4).sum()), "of", d.shape[0] * d.shape[1])
