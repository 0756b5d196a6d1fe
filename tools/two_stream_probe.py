"""Throughput of the hot-path step with ONE batch in flight (bench.py's loop) against TWO batches in flight on two HIP streams
(step k+1 is enqueued before step k's match counts are read back): do the MFMA-bound matching GEMM of one batch and the
latency-bound gather kernels of the other overlap?"""
import gc, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from casmtr_amd.pipeline import HotPath, HotPathConfig, make_synthetic_inputs

cfg = HotPathConfig.named(sys.argv[1] if len(sys.argv) > 1 else "4c")
model = HotPath(cfg).cuda()
inps = [make_synthetic_inputs(cfg, 8, "cuda", seed=1 + i) for i in range(2)]
N = 100
for i in range(4):
    model(inps[i % 2])
torch.cuda.synchronize()
gc.collect(); gc.freeze(); gc.disable()
t0 = time.perf_counter()
for i in range(N):
    model(inps[i % 2])
torch.cuda.synchronize()
one = (time.perf_counter() - t0) / N * 1e3
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for s in streams:
    s.wait_stream(torch.cuda.current_stream())


def enqueue(i):
    with torch.cuda.stream(streams[i % 2]):
        return model(inps[i % 2], finalize=False)


def finish(i, out):
    with torch.cuda.stream(streams[i % 2]):
        return model.finalize(out)


for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pend = enqueue(0)
    for i in range(N):
        nxt = enqueue(i + 1) if i + 1 < N else None
        finish(i, pend)
        pend = nxt
    torch.cuda.synchronize()
    two = (time.perf_counter() - t0) / N * 1e3
print(json.dumps({"config": cfg.name, "ms_per_step_one_in_flight": round(one, 3), "ms_per_step_two_in_flight": round(two, 3),
                  "pairs_per_s": [round(8e3 / one, 1), round(8e3 / two, 1)]}))
