"""CPU: the C-ABI library loads and exports every symbol include/casmtr_hip.h declares (no compute calls without a
GPU), the drop-in names resolve, host-side logic behaves, and the product never touches the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "casmtr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(casmtr_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported():
    from casmtr_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/casmtr_hip.h but not exported"
    # and the python binding knows every one of them
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert _lib.lib().casmtr_abi_version() == 8
    assert _lib.lib().casmtr_dual_softmax_ws_bytes(1, 10816, 10816) > 0


def test_library_is_gfx950_code_object():
    from casmtr_amd import _lib
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", _lib.LIB_PATH], capture_output=True, text=True)
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob, "library does not embed a gfx950 code object"
    assert out.returncode == 0


def test_missing_library_fails_loudly(monkeypatch):
    from casmtr_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcasmtr_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_ops_reject_host_tensors():
    from casmtr_amd import ops
    q = torch.zeros((1, 4, 4, 2, 32))
    with pytest.raises(RuntimeError, match="CUDA"):
        ops.qta_score_fwd(q, torch.zeros((1, 16, 2, 32)), torch.zeros((1, 4, 8, 2), dtype=torch.int64))
    with pytest.raises(RuntimeError, match="CUDA"):
        ops.window_score_fwd(torch.zeros(1, 16, 128), torch.zeros(1, 16, 128), torch.zeros((1, 16, 4), dtype=torch.int64))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "casmtr_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp")):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "libcasmtr_oracle" not in src and not re.search(r"#include\s*[<\"].*oracle", src), f"{f} links the oracle"
                assert not re.search(r"\borc_[a-z_]+\s*\(", src), f"{f} calls an oracle function"
    # importing the whole product must not pull the oracle in either
    code = "import sys; import casmtr_amd.pipeline, casmtr_amd.compat, casmtr_amd.dist; assert 'oracle' not in sys.modules"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)


def test_drop_in_names():
    import casmtr_amd.compat as compat
    compat.install()
    import fast_score_computation
    import score_computation_cuda
    import value_aggregation_cuda
    from cuda_imp.QuadTreeAttention.QuadtreeAttention.functions.quadtree_attention import (score_computation_op,
                                                                                             value_aggregation_op)
    from cuda_imp.QuadTreeAttention.QuadtreeAttention.modules.quadtree_attention import (CascadeQTAttB, QTAttA, QTAttB,
                                                                                           QTAttGuided)
    from src.model.functions.cascade_matching import CascadeMatching
    from src.model.functions.coarse_matching import CoarseMatching
    from src.model.functions.post_processing import PostProcess
    for m, fns in ((score_computation_cuda, ("score_forward", "score_backward")),
                   (value_aggregation_cuda, ("value_aggregation_forward", "value_aggregation_backward")),
                   (fast_score_computation, ("score_forward", "score_backward"))):
        for f in fns:
            assert callable(getattr(m, f))
    m = QTAttB(8, 32, 3, topks=[32, 16, 8])
    assert list(m.state_dict()) == ["weight"] and m.state_dict()["weight"].shape == (3,)
    ml = QTAttB(8, 32, 3, topks=[32, 16, 8], lepe=True)
    assert any(k.startswith("get_vs.0.") for k in ml.state_dict())
    assert list(CascadeQTAttB(4, 32, dilated=None).state_dict()) == [] and CascadeQTAttB(4, 32, None).dilated == 1
    assert list(QTAttGuided(8, 32, 3, topks=[16, 8, 8]).state_dict()) == ["weight"]
    assert list(QTAttA(8, 32, topks=[8, 4, 2]).state_dict()) == []   # variant A has no parameters (:8-22)
    with pytest.raises(NotImplementedError):
        PostProcess({"method": "sift"})
    assert PostProcess({"method": "maxpool_nms", "window_size": 5}).nms_window == 5
    assert PostProcess({"method": None}).nms_window == 0
    cfg = {"thr": 0.2, "border_rm": 0, "train_coarse_percent": 0.3, "train_pad_num_gt_min": 200,
           "match_type": "dual_softmax", "dsmax_temperature": 0.1}
    assert list(CoarseMatching(cfg).state_dict()) == []
    assert callable(score_computation_op) and callable(value_aggregation_op)
    cm = CascadeMatching({"thr": 0.2, "test_thr": 0.2, "pre_thr": [0.2], "border_rm": 2, "double_check": True,
                          "train_pad_num_gt_min": 200, "match_type": "softmax", "dsmax_temperature": 1.0},
                         {"propagation": "window", "dilated": 1, "post_config": {"method": "maxpool_nms", "window_size": 5}}, "4c")
    cm.train()
    with pytest.raises(NotImplementedError):
        cm.forward(None, None, None, None, {})


def test_layout_helpers_roundtrip():
    from casmtr_amd.modules.quadtree_attention import _quad_order, _raster_order
    x = torch.arange(2 * 6 * 8 * 3).float().view(2, 48, 3)
    qd = _quad_order(x, 6, 8)
    assert qd.shape == (2, 12, 4, 3)
    # child f of quad (qy,qx) is pixel (2qy + f//2, 2qx + f%2)
    assert torch.equal(qd[0, 5, 3], x[0, (2 * 1 + 1) * 8 + 2 * 1 + 1])
    assert torch.equal(_raster_order(qd, 6, 8), x)


def test_algorithmic_work_matches_survey():
    from casmtr_amd.pipeline import HotPathConfig, algorithmic_work
    w = algorithmic_work(HotPathConfig())
    assert abs(w["qta_bytes"] / 1e6 - 54.7) < 0.1 and abs(w["qta_flops"] / 1e9 - 1.531) < 0.01
    assert abs(w["coarse_flops"] / 1e9 - 59.90) < 0.01 and abs(w["cascade_bytes"] / 1e6 - 127.5) < 0.1
    assert abs(w["match_bytes"] / 1e6 - 193.5) < 0.1
    assert abs(w["total_bytes"] / 1e9 - 1.383) < 0.001 and abs(w["total_flops"] / 1e9 - 89.35) < 0.01


def test_valid_extents_and_border_masks():
    from casmtr_amd.matching.cascade_functions import mask_window_border, valid_extents
    m0 = torch.zeros(2, 6, 8, dtype=torch.bool)
    m0[0, :5, :7] = True
    m0[1] = True
    ext = valid_extents(m0, m0)
    assert ext.tolist() == [[5, 7, 5, 7], [6, 8, 6, 8]] and ext.dtype == torch.int32
    mask = torch.ones(1, 6, 8, dtype=torch.bool)
    idx = torch.zeros(1, 6, 8, 2, dtype=torch.long) + 3
    idx[0, 3, 3] = torch.tensor([0, 3])   # target row 0 < b
    out = mask_window_border(mask, idx, 2, False, 6, 8)
    assert not out[0, 3, 3] and out[0, 2, 2] and not out[0, 0, 4] and not out[0, 3, 7]


def test_compat_aliases_block_module():
    import sys
    import casmtr_amd.compat as compat
    compat.install()
    mod = sys.modules["src.model.modules.quadtree_attention"]
    from casmtr_amd.modules.quadtree_block import CascadeQuadtreeAttention, QuadtreeAttention
    assert mod.QuadtreeAttention is QuadtreeAttention and mod.CascadeQuadtreeAttention is CascadeQuadtreeAttention
    m = QuadtreeAttention(64, 2, [8, 4, 2], scale=3)
    assert sorted(m.state_dict()) == sorted(["q_proj.weight", "k_proj.weight", "v_proj.weight", "py_att.weight",
                                             "proj.weight", "proj.bias"])
    assert tuple(m.q_proj.weight.shape) == (64, 64, 1, 1)


def test_quad_kernels_isa_guard():
    """fine_quad.hip / cascade_quad.hip issue their LDS-DMA chunks from inline asm that writes M0 and is waited for with hand-counted
    vmcnt: the compiled kernels must not touch M0 elsewhere and must not spill (tools/check_quad_isa.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_quad_isa.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def _load_bench():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_host_tests", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_bench_audit_match_diff_names_causes():
    """bench.py's parity block prints WHY a match entry differs between the GPU chain and the oracle chain"""
    import numpy as np
    from types import SimpleNamespace
    bench = _load_bench()
    h, w = 8, 8
    st = SimpleNamespace(test_thr=0.2, pre_thr=[0.2], nms_window=5)
    cg = np.full((h * w,), 0.5, np.float32)
    co = cg.copy()
    cg[10], co[10] = 0.2000001, 0.1999999              # conf on the threshold: in the GPU list only
    co[27], co[28] = 0.9000001, 0.9000000              # two confidences of one NMS window within 2e-5: the winner flips
    cg[27], cg[28] = 0.9000000, 0.9000001
    pre = [(np.full((4 * 4,), 0.7, np.float32), (4, 4))]
    res = bench.audit_match_diff({(10, 3), (28, 5)}, {(27, 5)}, cg, co, (h, w), st, pre)
    by_i = {r["i"]: r for r in res}
    assert set(by_i) == {10, 27, 28}
    assert by_i[10]["in"] == "gpu only" and "test_thr" in by_i[10]["cause"][0]
    assert by_i[27]["in"] == "oracle only" and "NMS" in " ".join(by_i[27]["cause"]) and "NMS" in " ".join(by_i[28]["cause"])
    res = bench.audit_match_diff({(40, 1)}, set(), cg, co, (h, w), SimpleNamespace(test_thr=0.2, pre_thr=[0.2], nms_window=0), pre)
    assert res[0]["cause"] == ["UNEXPLAINED"], "an entry with no borderline comparison behind it is reported as such, not explained away"


def test_bench_live_traffic_parser(tmp_path, monkeypatch):
    """bench.measure_traffic_live: two rocprofv3 --pmc passes -> bytes per launch per scope with the gfx950 correction; a failing pass
    is reported, never raised (the bench line must survive a box without a usable rocprofv3)"""
    import subprocess
    bench = _load_bench()
    calls = []

    def fake_run(cmd, **kw):
        ctr = cmd[cmd.index("--pmc") + 1]
        d = cmd[cmd.index("-d") + 1]
        assert "--no-pmc" in cmd and "--no-extra" in cmd and kw.get("cwd") == "/tmp"
        os.makedirs(os.path.join(d, "host"), exist_ok=True)
        with open(os.path.join(d, "host", "1_counter_collection.csv"), "w") as f:
            f.write("Kernel_Name,Counter_Name,Counter_Value\n")
            for v in (100.0, 300.0):
                f.write(f'"void fine_quad_kernel<1, false, true>(FineQArgs)",{ctr},{v}\n')
            f.write(f'"void cascade_quad_kernel<false>(CasQArgs)",{ctr},1000.0\n')
            f.write(f'"void at::native::something()",{ctr},5.0\n')
        calls.append(ctr)
        return subprocess.CompletedProcess(cmd, 0, "", "")

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(bench.os.path, "exists", lambda p: True)
    out, note = bench.measure_traffic_live(["--config", "4c"])
    assert calls == ["FETCH_SIZE", "WRITE_SIZE"] and "measured in this run" in note
    assert out["qta_fine_level[lists<=64]"] == 2 * 200 * 1024 + 200 * 1024 and out["cascade_attn"] == 3 * 1000 * 1024
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: subprocess.CompletedProcess(cmd, 1, "", "boom"))
    out, note = bench.measure_traffic_live([])
    assert out == {} and "failed" in note
