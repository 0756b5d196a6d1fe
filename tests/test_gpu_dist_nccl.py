"""GPU: the RCCL code path of bench.py / casmtr_amd.dist on the one GPU a test box has.  Two ranks cannot share a device
under RCCL, so the process group is forced up with a single rank (CASMTR_FORCE_DIST=1): init_process_group("nccl"),
broadcast of the parameter buffer, barrier, counts all-gather, padded gather of the match lists and the max-over-ranks
all-reduce all run through RCCL exactly as they do at N > 1 (the N > 1 logic itself is covered by the gloo world-2 tests)."""
import json
import os
import subprocess
import sys

import pytest

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
