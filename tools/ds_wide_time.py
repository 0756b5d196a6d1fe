"""Split dual-softmax GEMM at the bench size: the 128 x 64-wave-tile kernel (ds_gemm16w_kernel) against the 64 x 64 one
(the default; CASMTR_DS_GEMM16_WIDE=1 selects the wide one), and the wide kernel without its epilogue / without its MFMAs (debug flags 4096 / 8192; timing only)."""
import os
import sys

os.environ["CASMTR_DEBUG_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from casmtr_amd import _lib, ops

B, h, C = 8, 104, 256
g = torch.Generator(device="cpu").manual_seed(0)
f0 = torch.randn((B, h * h, C), generator=g).cuda()
f1 = torch.randn((B, h * h, C), generator=g).cuda()
run = lambda: ops.dual_softmax(f0, f1, (h, h), (h, h), 0.1, 0.2, want_conf=False, gemm="split")


def scopes(n=8):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    _lib.prof_enable(True)
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    t = _lib.prof_read()
    _lib.prof_enable(False)
    return {k: round(v[0] / v[1], 4) for k, v in t.items() if "gemm" in k}


print("64 x 64 wave tiles:", scopes(), flush=True)
os.environ["CASMTR_DS_GEMM16_WST"] = "1"
print("64 x 64 wave tiles, 1 KB stores from the slabs:", scopes(), flush=True)
os.environ.pop("CASMTR_DS_GEMM16_WST")
print("64 x 64 wave tiles:", scopes(), flush=True)
os.environ["CASMTR_DS_GEMM16_WST"] = "1"
print("64 x 64 wave tiles, 1 KB stores from the slabs:", scopes(), flush=True)
os.environ.pop("CASMTR_DS_GEMM16_WST")
for rep in range(2):
    for st in ("3", "4"):
        os.environ["CASMTR_DS_GEMM16_STAGES"] = st
        print(f"64 x 64 wave tiles, CASMTR_DS_GEMM16_STAGES={st}:", scopes(), flush=True)
os.environ.pop("CASMTR_DS_GEMM16_STAGES")
os.environ["CASMTR_DS_GEMM16_WIDE"] = "1"
print("128 x 64 wave tiles + strips:", scopes(), flush=True)
for flags, name in ((4096, "no epilogue"), (8192, "no MFMAs"), (4096 + 8192, "DMA + LDS reads only")):
    _lib.lib().casmtr_debug_set(flags)
    print(f"128 x 64 wave tiles, {name}:", scopes(), flush=True)
_lib.lib().casmtr_debug_set(0)
