"""What does this box's HBM give to the simplest streaming kernels?  (reference points for the layout pass: read N + write N bytes)
python tools/hbm_probe.py -> GB/s of torch clone (read + write), fill (write), sum (read) at 350 MB and 1.06 GB."""
import torch

for mb in (350, 1060):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device="cuda")
    y = torch.empty_like(x)

    def t(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3

    tc, tf, ts = t(lambda: y.copy_(x)), t(lambda: y.fill_(1.0)), t(lambda: x.sum())
    print(f"{mb} MB: copy {2 * n * 4 / tc / 1e9:.0f} GB/s (read + write), fill {n * 4 / tf / 1e9:.0f} GB/s, sum {n * 4 / ts / 1e9:.0f} GB/s")
