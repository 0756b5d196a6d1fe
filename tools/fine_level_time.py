"""Times the two fine-level launches (K=128, K=64) in isolation with HIP events."""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from casmtr_amd import ops, _lib

B, H, C = 8, 8, 256
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device="cuda")
q2, k2, v2 = rn(B, 676, C), rn(B, 676, C), rn(B, 676, C)
q1, k1, v1 = rn(B, 2704, C), rn(B, 2704, C), rn(B, 2704, C)
q0, k0, v0 = rn(B, 10816, C), rn(B, 10816, C), rn(B, 10816, C)
l0 = ops.qta_coarse_level(q2, k2, v2, H, 32, w_level=0.3, want_message=False)
l1 = ops.qta_fine_level(q1, k1, v1, l0["topk_idx"], (52, 52), (52, 52), H, 16, w_level=0.3, acc_in=l0["acc"], want_message=False)
idx1 = l1["topk_idx"].clone()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tl1 = t(lambda: ops.qta_fine_level(q1, k1, v1, l0["topk_idx"], (52, 52), (52, 52), H, 16, w_level=0.3, acc_in=l0["acc"], want_message=False))
tl0 = t(lambda: ops.qta_fine_level(q0, k0, v0, idx1, (104, 104), (104, 104), H, 0, w_level=0.4, acc_in=l1["acc"], want_message=False))
print(f"STOP={os.environ.get('CASMTR_QUAD_STOP', '0')} VAR={os.environ.get('CASMTR_QUAD_VAR', '0')}  K=128 launch {tl1*1e3:.1f} us   K=64 launch {tl0*1e3:.1f} us")
