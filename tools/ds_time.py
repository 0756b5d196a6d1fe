"""Time the dual-softmax passes (exact fp32 MFMA vs f16 split) at the bench size: python tools/ds_time.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from casmtr_amd import ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
h, C = 104, 256
g = torch.Generator(device="cpu").manual_seed(0)
f0 = torch.randn((B, h * h, C), generator=g).cuda()
f1 = torch.randn((B, h * h, C), generator=g).cuda()
for masked in (False, True):
    m0 = valid = None
    if masked:
        m = torch.ones((B, h, h), dtype=torch.bool)
        m[:, 83:], m[:, :, 90:] = False, False
        m0 = m.reshape(B, -1).cuda()
        valid = torch.tensor([[83, 90, 83, 90]] * B, dtype=torch.int32).cuda()
    for gemm in ("exact", "split"):
        run = lambda: ops.dual_softmax(f0, f1, (h, h), (h, h), 0.1, 0.2, mask0=m0, mask1=m0, valid_hw=valid, want_conf=False, gemm=gemm)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            d = run()
        b.record()
        torch.cuda.synchronize()
        _lib.prof_enable(True)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        t = _lib.prof_read()
        _lib.prof_enable(False)
        print(f"masked={masked} gemm={gemm}: {a.elapsed_time(b) / 10:.3f} ms per call;", {k: round(v[0] / v[1], 4) for k, v in t.items()})
