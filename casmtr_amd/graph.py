"""HIP graphs for the hot path: a whole HotPath step (~100 kernel launches, every intermediate allocation) captured once and replayed.

The step has no data-dependent control flow on the host until its single read-back (HotPath.forward(finalize=False) enqueues
everything, finalize() reads the match counts), so it captures as it stands with torch.cuda.CUDAGraph -- the C ABI launches on the
current stream, which is the capturing stream.  Replaying removes the per-launch host work and the gaps between the small kernels
(tools/graph_probe.py, one MI355X, 8 CasMTR-4c pairs: 13.06 ms eager with a read-back per step, 12.37 ms per bare replay, 12.68 ms
with the lists of every step read back one step behind; CasMTR-2c: 21.6 / 21.5 ms -- its step is two thirds long kernels).

Inputs are the tensors the graph was captured on (copy new data into them, or let the producer write there).  Every replay
overwrites the graph's output buffers, so the few small tensors finalize() reads (match counts and the capacity-sized lists of
every stage, ~12 MB) are copied into one of `n_slots` snapshot sets behind each replay: step k's lists can be read while step k+1
runs.  Two ROCm 7.0 observations that shaped this: (1) hipMemsetAsync nodes are not reliably ordered against the kernel nodes around
them -- replays of a graph containing the dual-softmax workspace memset faulted after tens of steps, so the library clears that
workspace with a kernel (csrc/matching.hip); (2) HIP events recorded during capture cannot be read back afterwards
(hipEventSynchronize: hipErrorInvalidHandle), so the casmtr_prof_* per-kernel timing must stay off while capturing and bench.py's
live per-kernel numbers come from eager steps."""
from typing import Dict, List

import torch

# What finalize() reads of a stage's `_pending` entry (CoarseMatching.finalize / CascadeMatching.finalize): the device-side count and
# the capacity-sized lists.  These are ALWAYS snapshotted, whatever their size (at the 1/2 level a batch of 25+ pairs at 832x832
# has lists of more than 4 M entries).  Everything else in the entry (conf / similarity matrices, workspaces, dense per-token
# outputs) stays a reference to the graph's own buffer: valid until the next replay, never read by finalize().
_SNAP_KEYS = ("n", "b_ids", "i_ids", "j_ids", "mconf")


def _snap_struct(x, make, key=None):
    if torch.is_tensor(x):
        return make(x) if key in _SNAP_KEYS else x
    if isinstance(x, dict):
        return {k: _snap_struct(v, make, k) for k, v in x.items()}
    if isinstance(x, (tuple, list)):
        return type(x)(_snap_struct(v, make, key) for v in x)
    return x


def _copy_struct(dst, src):
    if torch.is_tensor(src):
        if dst is not src:
            dst.copy_(src, non_blocking=True)
    elif isinstance(src, dict):
        for k in src:
            _copy_struct(dst[k], src[k])
    elif isinstance(src, (tuple, list)):
        for d, s in zip(dst, src):
            _copy_struct(d, s)


class GraphedHotPath:
    def __init__(self, model, inp: Dict[str, object], n_slots: int = 2, warmup: int = 2):
        self.model, self.inp = model, inp
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):   # lazy one-time work (function attributes, occupancy queries, allocator growth) outside the capture
            for _ in range(warmup):
                model(inp, finalize=False)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = model(inp, finalize=False)
        torch.cuda.synchronize()
        # finalize() consumes a stage's `_pending` entry (raw capacity-sized lists + the device-side count)
        self._pending = {k: st["_pending"] for k, st in self.out["data"].items() if isinstance(st, dict) and "_pending" in st}
        self._slots: List[Dict[str, object]] = [{k: _snap_struct(p, torch.empty_like) for k, p in self._pending.items()} for _ in range(n_slots)]
        self._k = 0

    def enqueue(self) -> Dict[str, object]:
        """replay on the current stream -> a (not yet finalised) output dict whose lists live in the next snapshot slot; finalize() it
        before this slot comes round again (n_slots enqueues later).  `messages` and the dense per-token outputs are the graph's
        own buffers: valid until the next enqueue."""
        slot = self._slots[self._k % len(self._slots)]
        self._k += 1
        self.graph.replay()
        data = {}
        for k, v in self.out["data"].items():
            data[k] = dict(v) if isinstance(v, dict) else v
        for k, p in self._pending.items():
            _copy_struct(slot[k], p)
            data[k]["_pending"] = slot[k]
        return {"messages": self.out["messages"], "data": data}

    def finalize(self, out):
        return self.model.finalize(out)

    def step(self):
        return self.finalize(self.enqueue())
