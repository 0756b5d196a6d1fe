import sys, os
sys.path.insert(0, "/root/repo")
import torch
from casmtr_amd import ops, _lib
B, h, C = 8, 104, 256
g = torch.Generator(device="cpu").manual_seed(0)
f0 = torch.randn((B, h * h, C), generator=g).cuda(); f1 = torch.randn((B, h * h, C), generator=g).cuda()
run = lambda: ops.dual_softmax(f0, f1, (h, h), (h, h), 0.1, 0.2, want_conf=False, gemm="split")
for rep in range(3):
    for pr in ("0", "1", "2"):
        os.environ["CASMTR_DS_PRIO"] = pr
        for _ in range(3): run()
        torch.cuda.synchronize()
        _lib.prof_enable(True)
        for _ in range(10): run()
        torch.cuda.synchronize()
        t = _lib.prof_read(); _lib.prof_enable(False)
        print("prio", pr, round(t["dual_softmax_gemm"][0] / t["dual_softmax_gemm"][1], 4), flush=True)
