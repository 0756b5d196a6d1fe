import torch, time
import torch.nn.functional as F
torch.manual_seed(0)
for (B, nh, N, S) in [(16, 4, 43264, 676), (16, 8, 10816, 676)]:
    q = torch.randn(B, nh, N, 32, device="cuda"); k = torch.randn(B, nh, S, 32, device="cuda"); v = torch.randn(B, nh, S, 32, device="cuda")
    sc = 32 ** -0.5
    def a():
        return ((q @ k.transpose(-2, -1)) * sc).softmax(-1) @ v
    def b():
        return F.scaled_dot_product_attention(q, k, v, scale=sc)
    for f in (a, b):
        try:
            o = f(); torch.cuda.synchronize()
            t = time.time()
            for _ in range(3): o = f()
            torch.cuda.synchronize()
            print(f.__name__, (B, nh, N, S), f"{(time.time() - t) / 3 * 1e3:.2f} ms")
        except Exception as e:
            print(f.__name__, "failed", str(e)[:200])
    print("max diff", float((a() - b()).abs().max()))
