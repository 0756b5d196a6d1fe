"""One process per GPU.  Image pairs are independent, so the data path has NO collective; torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm, "gloo" on CPU for tests) carries only
  * broadcast_parameters(): the start-up broadcast of the flat fp32 parameter buffer from rank 0, and
  * gather_matches(): the end-of-step gatherv of the match lists to rank 0 (per-rank counts all-gather, then a
    padded gather of [M,5] fp32 (mkpts0, mkpts1, mconf) and [M] int64 (m_bids, offset to global pair ids)).
This replaces the role of the reference's detectron2-style pickled gathers on a gloo side group
(src/utils/comm.py:84-220) for the inference path.
"""
import os

import torch
import torch.distributed as dist


# CASMTR_FORCE_DIST=1: initialise the process group and run every collective even with a single rank -- lets a 1-GPU box
# exercise the RCCL code path (tests/test_gpu_dist_nccl.py); two ranks cannot share one device under RCCL.
_FORCE = os.environ.get("CASMTR_FORCE_DIST", "0") == "1"


def init_from_env(backend=None):
    """-> (rank, world, local_rank); initialises the default group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or _FORCE) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            # CASMTR_DIST_BACKEND=gloo: the collectives run over gloo on host copies although the data lives on GPUs -- lets several
            # ranks SHARE one device (RCCL refuses that), i.e. a one-GPU box can run the real N > 1 flow end to end (tests)
            backend = os.environ.get("CASMTR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def is_dist():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def barrier():
    if is_dist():
        dist.barrier()


def finalize():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def max_over_ranks(x: float) -> float:
    if not is_dist():
        return x
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_range(n_items: int, rank: int, world: int):
    """contiguous block partition of n_items pairs (what DistributedSampler(shuffle=False) gives the reference,
    src/lightning/data.py:316, up to interleaving)."""
    per, rem = divmod(n_items, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """One flat fp32 buffer, one broadcast (the hot path owns 3 floats; a full CasMTR-4c is 56 MB -- still one call)."""
    params = [p for p in module.parameters()] + [b for b in module.buffers()]
    if not is_dist() or not params:
        return
    flat = torch.cat([p.detach().reshape(-1).float() for p in params])
    if dist.get_backend() == "gloo" and flat.is_cuda:
        host = flat.cpu()
        dist.broadcast(host, src=src)
        flat = host.to(flat.device)
    else:
        dist.broadcast(flat, src=src)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))
            off += n


def gather_matches(out, pairs_per_rank=None, dst: int = 0, pair_offset=None):
    """out: dict with m_bids [M] int64, mkpts0/mkpts1 [M,2], mconf [M].  Returns on rank `dst` a dict with the
    concatenated lists (m_bids offset to global pair ids) and n_total; None elsewhere.  Single process: passthrough.
    Global pair id = local id + pair_offset (this rank's shard_range lower bound -- correct for uneven shards too);
    pairs_per_rank is the equal-shard shorthand for pair_offset = rank * pairs_per_rank."""
    mk = torch.cat([out["mkpts0"].float(), out["mkpts1"].float(), out["mconf"].float()[:, None]], dim=1)  # [M,5]
    bids = out["m_bids"]
    if not is_dist():
        return {"mk": mk, "m_bids": bids, "n_total": int(mk.shape[0]), "counts": [int(mk.shape[0])]}
    world, rank = dist.get_world_size(), dist.get_rank()
    if dist.get_backend() == "gloo" and mk.is_cuda:   # gloo gathers host tensors (see init_from_env)
        mk, bids = mk.cpu(), bids.cpu()
    dev = mk.device
    cnt = torch.tensor([mk.shape[0]], dtype=torch.int64, device=dev)
    all_cnt = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_cnt, cnt)
    counts = all_cnt.tolist()   # ONE device->host read-back for all ranks' counts
    mmax = max(max(counts), 1)
    pad_mk = torch.zeros((mmax, 5), dtype=torch.float32, device=dev)
    pad_b = torch.zeros((mmax,), dtype=torch.int64, device=dev)
    pad_mk[: mk.shape[0]] = mk
    if pair_offset is None and pairs_per_rank is not None:
        pair_offset = rank * pairs_per_rank
    if pair_offset:
        bids = bids + int(pair_offset)
    pad_b[: bids.shape[0]] = bids
    if rank == dst:
        g_mk = [torch.empty_like(pad_mk) for _ in range(world)]
        g_b = [torch.empty_like(pad_b) for _ in range(world)]
        dist.gather(pad_mk, g_mk, dst=dst)
        dist.gather(pad_b, g_b, dst=dst)
        return {"mk": torch.cat([g[:c] for g, c in zip(g_mk, counts)]),
                "m_bids": torch.cat([g[:c] for g, c in zip(g_b, counts)]), "n_total": sum(counts), "counts": counts}
    dist.gather(pad_mk, None, dst=dst)
    dist.gather(pad_b, None, dst=dst)
    return None


class MatchGatherer:
    """gather_matches() every `every` steps instead of every step: the lists of the steps in between stay on their rank (device
    tensors, tens of KB each) and travel in ONE exchange -- one counts all-gather, one host read-back and two padded gathers per
    `every` steps.  Pair ids in the gathered m_bids are step * pairs_per_step + global pair id, steps counted from the last flush.
    every = 1 is the reference's behaviour (results leave the rank after every batch).  Single process: plain passthrough.
    add() and flush() are COLLECTIVE: every rank calls them the same number of times (a rank that holds lists while another one
    does not would deadlock the gather), and lists still held when the object is dropped are lost -- call flush() last."""

    MAX_HELD_STEPS = 4096   # a window larger than this (every = "never": one final gather) still exchanges after this many steps, so that a
                            # long-running caller does not hold every step's lists forever (identical on all ranks: stays collective)

    def __init__(self, every=1, pair_offset=0, pairs_per_step=0, dst=0):
        self.every, self.pair_offset, self.pairs_per_step, self.dst = max(1, int(every)), int(pair_offset), int(pairs_per_step), dst
        self.every = min(self.every, self.MAX_HELD_STEPS)
        if self.every > 1 and self.pairs_per_step <= 0:
            raise ValueError("MatchGatherer(every > 1) needs pairs_per_step > 0: the held steps' pair ids would collapse onto each other")
        self._held = []

    def add(self, out):
        """-> what gather_matches returns when this call triggered an exchange (rank dst: dict, others: None), else None"""
        if not is_dist() or self.every == 1:
            return gather_matches(out, dst=self.dst, pair_offset=self.pair_offset)
        self._held.append({k: out[k] for k in ("m_bids", "mkpts0", "mkpts1", "mconf")})
        return self.flush() if len(self._held) >= self.every else None

    def flush(self):
        if not self._held:
            return None
        held, self._held = self._held, []
        both = {k: torch.cat([h[k] for h in held]) for k in ("mkpts0", "mkpts1", "mconf")}
        both["m_bids"] = torch.cat([h["m_bids"] + i * self.pairs_per_step for i, h in enumerate(held)])
        return gather_matches(both, dst=self.dst, pair_offset=self.pair_offset)
