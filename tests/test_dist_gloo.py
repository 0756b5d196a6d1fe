"""CPU: the N>1 path (sharding, parameter broadcast, gatherv of matches) with world_size 2 over gloo."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from casmtr_amd import dist as cdist
    r, w, _ = cdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and cdist.is_dist()
    # start-up broadcast of the parameter buffer
    m = torch.nn.Linear(3, 2)
    with torch.no_grad():
        for p in m.parameters():
            p.fill_(float(rank + 1))
    cdist.broadcast_parameters(m)
    assert all(torch.all(p == 1.0) for p in m.parameters())
    # shard 5 pairs over 2 ranks: contiguous blocks covering everything exactly once
    lo, hi = cdist.shard_range(5, rank, world)
    # ragged gather, including an empty rank
    M = 0 if rank == 1 else 4
    out = {"m_bids": torch.arange(M) % 2, "mkpts0": torch.full((M, 2), float(rank)), "mkpts1": torch.full((M, 2), 7.0),
           "mconf": torch.linspace(0, 1, M) if M else torch.zeros(0)}
    res = cdist.gather_matches(out, pairs_per_rank=2)
    # uneven shards (5 pairs over 2 ranks = 3 + 2): global pair ids come from the shard's lower bound, not rank * per_rank
    out2 = {"m_bids": torch.arange(hi - lo), "mkpts0": torch.zeros((hi - lo, 2)), "mkpts1": torch.zeros((hi - lo, 2)),
            "mconf": torch.ones(hi - lo)}
    res2 = cdist.gather_matches(out2, pair_offset=lo)
    # --gather-every 3: two steps stay on the rank, the third triggers ONE exchange carrying all three
    g = cdist.MatchGatherer(every=3, pair_offset=lo, pairs_per_step=5)
    held = []
    for stepi in range(3):
        held.append(g.add({"m_bids": torch.arange(hi - lo), "mkpts0": torch.full((hi - lo, 2), float(stepi)),
                           "mkpts1": torch.zeros((hi - lo, 2)), "mconf": torch.ones(hi - lo)}))
    assert held[0] is None and held[1] is None and g.flush() is None
    assert (held[2] is not None) == (rank == 0)
    t = cdist.max_over_ranks(float(rank))
    cdist.barrier()
    if rank == 0:
        q.put(dict(range=(lo, hi), n=res["n_total"], counts=res["counts"], bids=res["m_bids"].tolist(), shape=tuple(res["mk"].shape), tmax=t,
                   uneven_bids=res2["m_bids"].tolist(), every3_bids=held[2]["m_bids"].tolist(), every3_step=held[2]["mk"][:, 0].tolist()))
    else:
        assert res is None
        q.put(dict(range=(lo, hi)))
    cdist.finalize()


def test_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ranges = sorted(o["range"] for o in outs)
    assert ranges == [(0, 3), (3, 5)]
    r0 = next(o for o in outs if "n" in o)
    assert r0["n"] == 4 and r0["counts"] == [4, 0] and r0["shape"] == (4, 5) and r0["bids"] == [0, 1, 0, 1] and r0["tmax"] == 1.0
    assert r0["uneven_bids"] == [0, 1, 2, 3, 4], "every pair of an uneven partition keeps its own global id"
    # MatchGatherer(every=3): rank 0's three steps, then rank 1's; ids = step * pairs_per_step + global pair id
    assert r0["every3_bids"] == [0, 1, 2, 5, 6, 7, 10, 11, 12, 3, 4, 8, 9, 13, 14]
    assert r0["every3_step"] == [0., 0., 0., 1., 1., 1., 2., 2., 2., 0., 0., 1., 1., 2., 2.]


def test_single_process_passthrough():
    from casmtr_amd import dist as cdist
    assert not cdist.is_dist() and cdist.max_over_ranks(2.5) == 2.5
    out = {"m_bids": torch.zeros(3, dtype=torch.long), "mkpts0": torch.ones(3, 2), "mkpts1": torch.ones(3, 2), "mconf": torch.ones(3)}
    res = cdist.gather_matches(out)
    assert res["n_total"] == 3 and res["mk"].shape == (3, 5)
    assert cdist.shard_range(32, 3, 8) == (12, 16)


def test_match_gatherer_needs_pairs_per_step():
    """holding steps back without pairs_per_step would fold every held step onto the same global pair ids"""
    import pytest
    from casmtr_amd import dist as cdist
    with pytest.raises(ValueError, match="pairs_per_step"):
        cdist.MatchGatherer(every=4)
    cdist.MatchGatherer(every=1)                       # the per-batch exchange needs none
    cdist.MatchGatherer(every=4, pairs_per_step=8)


# ----------------------------------------------------------------------------------------------------------------------------
# bench.py's own N > 1 code path, end to end, under gloo: `python -m torch.distributed.run ... bench.py --gpus 2
# --total-pairs 5 --mock-hotpath` runs main()'s sharding, parameter broadcast, MatchGatherer and timed loop on two CPU ranks; only the
# kernels are replaced by deterministic synthetic match lists (bench.mock_lists: empty for some (pair, step), 20 000 entries for pair 3).
def _expected_exchanges(total, world, steps, warmup, every):
    """what rank 0 must receive, recomputed independently from bench.mock_lists"""
    import importlib.util
    import os as _os
    root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_mock", _os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from casmtr_amd import dist as cdist
    ranges = [cdist.shard_range(total, r, world) for r in range(world)]
    window = every if every > 0 else 1 << 30
    out, held = [], []

    def flush():
        if not held:
            return
        counts, sb, sc = [], 0, 0.0
        for lo, hi in ranges:                      # rank-major, then held step, then pair: gather_matches' concatenation order
            c = 0
            for i, step in enumerate(held):
                for g in range(lo, hi):
                    n, code = bench.mock_lists(g, step)
                    c += n
                    sb += n * (g + (i * total if window > 1 else 0))
                    sc += n * code
            counts.append(c)
        out.append({"n_total": sum(counts), "counts": counts, "sum_m_bids": sb, "sum_code": sc, "mk_shape": [sum(counts), 5]})
        del held[:]

    for step in range(warmup, warmup + steps):
        held.append(step)
        if len(held) >= window:
            flush()
    flush()
    return out


def _run_mock_bench(every, total=5, world=2, steps=7, warmup=2, launcher=True, extra=(), env=None):
    """launcher=True: under `python -m torch.distributed.run` (the driver's documented N > 1 form); False: `python bench.py --gpus N`
    as a plain process, which must re-execute itself under the launcher (bench.self_launch_cmd)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup",
            str(warmup), "--total-pairs", str(total), "--gather-every", str(every), "--mock-hotpath", *extra]
    env = dict(os.environ, OMP_NUM_THREADS="1", **(env or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):   # a plain process is one that no launcher has prepared
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line, from rank 0"
    return json.loads(lines[0])


import pytest  # noqa: E402


@pytest.mark.parametrize("every", [0, 1, 3])
def test_bench_main_under_gloo_world2(every):
    """--gather-every 0 (one final exchange inside the timed region), 1 (per step) and 3 (windows of 3, last one partial)"""
    steps, warmup, total = 7, 2, 5
    d = _run_mock_bench(every, total=total, steps=steps, warmup=warmup)
    check_mock_line(d, every, steps, warmup, total)


def test_bench_self_launches_without_torchrun():
    """`python bench.py --gpus 2 ...` with no WORLD_SIZE in the environment (how a driver that does not know about launchers starts an
    N > 1 run) becomes two ranks under torch.distributed.run by itself and still prints exactly one JSON line"""
    steps, warmup, total = 7, 2, 5
    d = _run_mock_bench(3, total=total, steps=steps, warmup=warmup, launcher=False)
    assert d["backend"] == "gloo"
    check_mock_line(d, 3, steps, warmup, total)


def test_self_launch_command_line():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_cmd", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.self_launch_cmd(8, ["--gpus", "8", "--steps", "20", "--warmup", "10"], port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-7] == os.path.join(root, "bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "10"]
    port = int(bench.self_launch_cmd(2, [])[9])   # no port given: a free one is picked
    assert 1024 < port < 65536


def check_mock_line(d, every, steps, warmup, total, world=2):
    """rank 0's line of a --mock-hotpath run against the lists recomputed here (shared with tests/test_gpu_dist_nccl.py)"""
    from casmtr_amd import dist as cdist
    lo, hi = cdist.shard_range(total, 0, world)
    assert d["mock"] and d["n_gpus"] == world and d["scaling"] == "strong" and d["pairs_this_rank"] == hi - lo and d["first_timed_step"] == warmup
    want = _expected_exchanges(total, world, steps, warmup, every)
    assert len(d["exchanges"]) == len(want) == (1 if every == 0 else -(-steps // every))
    for got, w in zip(d["exchanges"], want):
        assert got["counts"] == w["counts"] and got["n_total"] == w["n_total"] and got["mk_shape"] == w["mk_shape"]
        assert got["sum_m_bids"] == w["sum_m_bids"], "global pair ids (shard offset + held-step offset)"
        assert abs(got["sum_code"] - w["sum_code"]) < 1e-3 * max(1.0, abs(w["sum_code"]))
    assert max(max(g["counts"]) for g in d["exchanges"]) >= 20000, "the large list travelled"


def test_match_gatherer_window_is_bounded():
    """every = 'never' (bench's one final gather) still exchanges after MAX_HELD_STEPS steps: a service loop cannot hold lists forever"""
    from casmtr_amd import dist as cdist
    g = cdist.MatchGatherer(every=1 << 30, pairs_per_step=8)
    assert g.every == cdist.MatchGatherer.MAX_HELD_STEPS
