import os, sys, numpy as np, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_model_harness as T
z = np.load(T.FIX); fx = {k: z[k] for k in z.files}
m = T._model(fx, "cuda")
f8 = torch.from_numpy(fx["f8"]).cuda().float()
from casmtr_amd.modules import quadtree_block as qb
outs = {}
for route in ("tokens", "quads"):
    os.environ["CASMTR_CALLER_LAYOUT"] = route
    rec = []
    hooks = [blk.register_forward_hook(lambda mod, i, o, rec=rec: rec.append(o.detach().clone())) for blk in m.modules() if isinstance(blk, qb.QuadtreeAttention)]
    with torch.no_grad():
        t0, t1 = m.loftr_coarse_8c(m.pos_encoding_8c(f8[:1]), m.pos_encoding_8c(f8[1:]))
    for h in hooks: h.remove()
    outs[route] = (torch.cat([t0, t1]), rec)
a, b = outs["tokens"], outs["quads"]
print("final max diff", float((a[0] - b[0]).abs().max()), "max abs", float(a[0].abs().max()))
for i, (x, y) in enumerate(zip(a[1], b[1])):
    d = (x - y).abs()
    print("attn call", i, tuple(x.shape), "max diff", float(d.max()), "max abs", float(x.abs().max()), "tokens > 1e-4:", int((d.amax(-1) > 1e-4).sum()))
ref = torch.from_numpy(fx["t8"].astype(np.float32))
for route in ("tokens", "quads"):
    x = outs[route][0].detach().float().cpu()
    err = (x - ref).abs() / (1.0 + ref.abs())
    print(route, "vs reference fixture: within 3e-3:", float((err <= 3e-3).float().mean()), "within 1e-2:", float((err <= 1e-2).float().mean()), "max", float(err.max()),
          "tokens with any element > 3e-3:", int((err.amax(-1) > 3e-3).sum()), "of", err.shape[0] * err.shape[1])
