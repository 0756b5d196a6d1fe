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
res = {"config": cfg.name, "ms_per_step_1_in_flight": round(one, 3)}
for NS in (2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(NS)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())

    def enqueue(i):
        with torch.cuda.stream(streams[i % NS]):
            return model(inps[i % 2], finalize=False)

    def finish(i, out):
        with torch.cuda.stream(streams[i % NS]):
            return model.finalize(out)

    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pend = [enqueue(i) for i in range(NS - 1)]
        for i in range(N):
            if i + NS - 1 < N:
                pend.append(enqueue(i + NS - 1))
            finish(i, pend.pop(0))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / N * 1e3
    res[f"ms_per_step_{NS}_in_flight"] = round(ms, 3)
print(json.dumps(res))
