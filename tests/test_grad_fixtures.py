"""the backward kernels against gradients produced by the REFERENCE python
(tests/golden/gen_golden.py::gen_grads runs the reference's own autograd.Functions and modules backward; the three
extension entry points it calls are torch-autograd derivatives of the authors' python equivalents).

CPU: the oracle's backward restatements are pinned to those fixtures.  GPU: the HIP backward kernels and the composed
(autograd) path of QTAttB / CascadeQTAttB are compared with the same fixtures and with the oracle.

Tolerance: gradients are fp32 sums of up to K*4 (query side) or of a data-dependent number (key / value side, atomics on
the GPU) of products; the bar is 1e-5 of the tensor's largest magnitude (+1e-6 absolute) -- rounding-order noise only.
"""
import os

import numpy as np
import pytest
import torch

import oracle
from golden_inputs import CASES, GRAD_CASES, checksum, make_grad_inputs, make_inputs
from parity_utils import GOLD

DEV = "cuda:0"


def _load(group, name):
    z = np.load(os.path.join(GOLD, f"grads_{group}_{name}.npz"))
    return {k: z[k] for k in z.files}


def _same_crc(stored, arrays):
    return (int(stored[0]) & 0xFFFFFFFF) == int(checksum(arrays)[0])   # fixtures store integers as int32


def _close(a, b, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    tol = 1e-5 * max(1.0, np.abs(b).max()) + 1e-6
    err = np.abs(a - b).max()
    assert err <= tol, f"{what}: max abs err {err:.3e} > {tol:.3e}"


def _ops_case(name):
    inp, gi, g = make_inputs("ops", name), make_grad_inputs("ops", name), _load("ops", name)
    assert _same_crc(g["checksum"], inp) and _same_crc(g["gchecksum"], gi), "RNG drift"
    B, N1, _, K, H = gi["agg_score"].shape
    flat = dict(idx5=np.repeat(inp["idx"][:, :, None], 4, axis=2).reshape(B, N1 * 4, K, H),
                agg_score=gi["agg_score"].reshape(B, N1 * 4, K, H), g_msg=gi["g_msg"].reshape(B, N1 * 4, H, -1))
    return inp, gi, g, flat


@pytest.mark.parametrize("name", GRAD_CASES["ops"])
def test_oracle_backward_vs_reference(name):
    inp, gi, g, fl = _ops_case(name)
    dq, dk = oracle.qta_score_bwd(gi["g_score"], inp["q"], inp["key"], inp["idx"])
    _close(dq, g["score_dq"], "score_backward dq (score_computation_kernal.cu:94-143)")
    _close(dk, g["score_dkey"], "score_backward dkey (:145-184)")
    gs, gv = oracle.qta_value_agg_bwd(fl["g_msg"], fl["agg_score"], inp["value"], fl["idx5"])
    _close(gs.reshape(g["agg_dscore"].shape), g["agg_dscore"], "value_aggregation_backward grad_score (value_aggregation_kernel.cu:55-86)")
    _close(gv, g["agg_dvalue"], "value_aggregation_backward grad_value")
    dq, dk = oracle.window_score_bwd(gi["g_window"], inp["wq"], inp["wkey"], inp["widx"])
    _close(dq, g["window_dq"], "score_cuda backward dq (score_computation_kernel.cu:67-123)")
    _close(dk, g["window_dkey"], "score_cuda backward dkey")


def test_reference_modules_agree_on_gradients():
    """the fixture generator ran the reference's pure-torch QTAttB and its CUDA-path QTAttB (over the stub ops): same grads"""
    for name in GRAD_CASES["qtattb"]:
        assert float(_load("qtattb", name)["smart_vs_cuda_maxabs"][0]) < 1e-5


# ------------------------------------------------------------------------------------------------------- GPU
def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", GRAD_CASES["ops"])
def test_hip_backward_vs_reference(name):
    from casmtr_amd import ops
    inp, gi, g, fl = _ops_case(name)
    dq, dk = ops.qta_score_bwd(T(gi["g_score"]), T(inp["q"]), T(inp["key"]), T(inp["idx"]))
    _close(N(dq), g["score_dq"], "HIP score dq"), _close(N(dk), g["score_dkey"], "HIP score dkey")
    gs, gv = torch.zeros(fl["agg_score"].shape, device=DEV), torch.zeros(inp["value"].shape, device=DEV)
    ops.qta_value_agg_bwd(T(fl["g_msg"]), T(fl["agg_score"]), T(inp["value"]), T(fl["idx5"]), gs, gv)
    _close(N(gs).reshape(g["agg_dscore"].shape), g["agg_dscore"], "HIP agg grad_score"), _close(N(gv), g["agg_dvalue"], "HIP agg grad_value")
    dq, dk = ops.window_score_bwd(T(gi["g_window"]), T(inp["wq"]), T(inp["wkey"]), T(inp["widx"]))
    _close(N(dq), g["window_dq"], "HIP window dq"), _close(N(dk), g["window_dkey"], "HIP window dkey")


@pytest.mark.gpu
@pytest.mark.parametrize("name", GRAD_CASES["ops"])
def test_autograd_functions_vs_reference(name):
    """the drop-in autograd.Functions (functions/quadtree_attention.py, cascade_functions.py) end to end"""
    from casmtr_amd.functions.quadtree_attention import score_computation_op, value_aggregation_op
    from casmtr_amd.matching.cascade_functions import ScoreComputation
    inp, gi, g, _ = _ops_case(name)
    q, key = T(inp["q"]).requires_grad_(True), T(inp["key"]).requires_grad_(True)
    score_computation_op(q, key, T(inp["idx"])).backward(T(gi["g_score"]))
    _close(N(q.grad), g["score_dq"], "score_computation_op dq"), _close(N(key.grad), g["score_dkey"], "score_computation_op dkey")
    sc, val = T(gi["agg_score"]).requires_grad_(True), T(inp["value"]).requires_grad_(True)
    idx5 = T(inp["idx"]).unsqueeze(2).repeat(1, 1, 4, 1, 1)
    value_aggregation_op(sc, val, idx5).backward(T(gi["g_msg"]))
    _close(N(sc.grad), g["agg_dscore"], "value_aggregation_op grad_score"), _close(N(val.grad), g["agg_dvalue"], "value_aggregation_op grad_value")
    wq, wk = T(inp["wq"]).requires_grad_(True), T(inp["wkey"]).requires_grad_(True)
    ScoreComputation.apply(wq, wk, T(inp["widx"])).backward(T(gi["g_window"]))
    _close(N(wq.grad), g["window_dq"], "ScoreComputation dq"), _close(N(wk.grad), g["window_dkey"], "ScoreComputation dkey")


@pytest.mark.gpu
@pytest.mark.parametrize("name", GRAD_CASES["qtattb"])
def test_qtattb_backward_vs_reference(name):
    from casmtr_amd.modules.quadtree_attention import QTAttB
    cfg = CASES["qtattb"][name]
    inp, gi, g = make_inputs("qtattb", name), make_grad_inputs("qtattb", name), _load("qtattb", name)
    assert _same_crc(g["checksum"], inp), "RNG drift"
    m = QTAttB(cfg["nhead"], cfg["D"], scale=3, topks=cfg["topks"]).to(DEV)
    with torch.no_grad():
        m.weight.copy_(T(inp["weight"]))
    qs, ks, vs = ([T(x).requires_grad_(True) for x in inp[n]] for n in ("queries", "keys", "values"))
    m(qs, ks, vs).backward(T(gi["g_final"]))
    for n, ts in (("q", qs), ("k", ks), ("v", vs)):
        for lv, t in enumerate(ts):
            _close(N(t.grad), g[f"d{n}{lv}"], f"QTAttB d{n} level {lv}")
    _close(N(m.weight.grad), g["dweight"], "QTAttB dweight")


@pytest.mark.gpu
@pytest.mark.parametrize("name", GRAD_CASES["cascade_attn"])
def test_cascade_backward_vs_reference(name):
    from casmtr_amd import ops
    from casmtr_amd.modules.quadtree_attention import CascadeQTAttB
    cfg = CASES["cascade_attn"][name]
    inp, gi, g = make_inputs("cascade_attn", name), make_grad_inputs("cascade_attn", name), _load("cascade_attn", name)
    assert _same_crc(g["checksum"], inp), "RNG drift"
    hc, wc = cfg["coarse_hw"]
    tp = ops.window_warp_idx(T(inp["coarse_idx"]), hc, wc, cfg["ws"])
    m = CascadeQTAttB(cfg["nhead"], cfg["D"], dilated=1).to(DEV)
    q, k, v = (T(inp[n]).requires_grad_(True) for n in "qkv")
    rel = T(inp["rel_pos"]).requires_grad_(True) if cfg.get("rel_pos") else None
    msg, _ = m(q, k, v, tp, rel)
    msg.backward(T(gi["g_message"]))
    _close(N(q.grad), g["dq"], "CascadeQTAttB dq"), _close(N(k.grad), g["dk"], "CascadeQTAttB dk"), _close(N(v.grad), g["dv"], "CascadeQTAttB dv")
    if rel is not None:
        _close(N(rel.grad), g["drel_pos"], "CascadeQTAttB drel_pos")
