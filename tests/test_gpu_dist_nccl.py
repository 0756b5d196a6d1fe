"""GPU: the RCCL code path of bench.py / casmtr_amd.dist on the one GPU a test box has.  Two ranks cannot share a device
under RCCL, so the process group is forced up with a single rank (CASMTR_FORCE_DIST=1): init_process_group("nccl"),
broadcast of the parameter buffer, barrier, counts all-gather, padded gather of the match lists and the max-over-ranks
all-reduce all run through RCCL exactly as they do at N > 1 (the N > 1 logic itself is covered by the gloo world-2 tests)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_under_torchrun_with_rccl_group():
    env = dict(os.environ, CASMTR_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["matches_last_step"] > 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: two ranks cannot share a device under RCCL")
@pytest.mark.parametrize("every", [0, 1, 3])
def test_two_rank_rccl(every):
    """The first box with two GPUs runs the REAL two-rank RCCL exchange: `python bench.py --gpus 2 --mock-hotpath --mock-device cuda`
    (self-launched, no torchrun on the command line) shards 5 pairs 3 + 2, broadcasts the parameter buffer, and gathers uneven, empty
    and 20 000-entry device-resident lists to rank 0 (dist.gather to a non-dst rank, MatchGatherer.flush) at --gather-every 0 / 1 / 3;
    the gathered lists must equal what a single process computes for all pairs (tests/test_dist_gloo.py::check_mock_line)."""
    from test_dist_gloo import _run_mock_bench, check_mock_line
    steps, warmup, total = 7, 2, 5
    d = _run_mock_bench(every, total=total, steps=steps, warmup=warmup, launcher=False, extra=("--mock-device", "cuda"),
                        env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["backend"] == "nccl" and d["device"].startswith("cuda")
    check_mock_line(d, every, steps, warmup, total)


def test_mock_device_cuda_single_rank_forced_rccl():
    """what a 1-GPU box can check of the same code: the mock lists as DEVICE tensors through a forced one-rank RCCL group"""
    env = dict(os.environ, CASMTR_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
           "--total-pairs", "5", "--gather-every", "3", "--mock-hotpath", "--mock-device", "cuda"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "nccl" and d["device"] == "cuda:0" and len(d["exchanges"]) == 2
    from test_dist_gloo import _expected_exchanges
    want = _expected_exchanges(5, 1, 4, 1, 3)
    assert [g["counts"] for g in d["exchanges"]] == [w["counts"] for w in want]
    assert [g["sum_m_bids"] for g in d["exchanges"]] == [w["sum_m_bids"] for w in want]


def _bench_line(extra, env=None, timeout=900):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "2", "--batch", "1", "--no-extra", "--no-cpu-baseline"] + extra
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CASMTR_FORCE_DIST"):
        e.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_two_ranks_share_the_gpu_over_gloo_with_the_real_hot_path():
    """The whole N > 1 flow with the REAL kernels on a one-GPU box: `python bench.py --gpus 2` (self-launched) with
    CASMTR_DIST_BACKEND=gloo puts both ranks on cuda:0 (RCCL would refuse to share a device; gloo carries host copies): each rank runs
    its own pair through the HIP hot path, rank 0's parameters are broadcast, the match lists are gathered to rank 0 after every step.
    Each rank's list must be exactly what a single process computes for that rank's inputs (seed 1234 + rank) with rank 0's parameters."""
    two = _bench_line(["--gpus", "2", "--gather-every", "1"], env={"CASMTR_DIST_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["config"]["pairs_per_step"] == 2 and two["scaling"] == "weak"
    per_rank = two["config"]["matches_gathered_per_rank"]
    assert len(per_rank) == 2 and sum(per_rank) == two["config"]["matches_gathered"] and min(per_rank) > 100
    for r in (0, 1):
        one = _bench_line(["--gpus", "1", "--seed-offset", str(r)] if r else ["--gpus", "1"])
        assert one["config"]["matches_last_step"] == per_rank[r], f"rank {r}: gathered {per_rank[r]}, single process {one['config']['matches_last_step']}"
