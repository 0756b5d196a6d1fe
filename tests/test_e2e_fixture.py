"""BASELINE configs[0] as a parity case: tensors captured from the reference's whole CasMTR-4c model (random init) on the
london_bridge demo pair (tests/golden/gen_golden_e2e.py), fed to the oracle (CPU) and to the HIP path (GPU).
Real-image activations are spatially coherent and contain near-flat regions: the near-tie audit is exercised for real."""
import os

import numpy as np
import pytest
import torch

import oracle
from parity_utils import GOLD, assert_close, audit_index_mismatches, audit_topk_mismatches, dot_score_fn, match_set

TOL = 1e-4
Z = np.load(os.path.join(GOLD, "e2e_london_bridge.npz"))
F32 = lambda k: Z[k].astype(np.float32)
I64 = lambda k: Z[k].astype(np.int64)
H4, W4, H8, W8, HI, WI = (int(x) for x in Z["meta"])


def _tok(x):
    B, C, h, w = x.shape
    return np.ascontiguousarray(x.transpose(0, 2, 3, 1).reshape(B, h * w, C))


def _qta_logit_fn(q_nchw, k_nchw, nhead=8):
    """float64 logits of (query l, head h) against selected keys, for audit_topk_mismatches"""
    q, k = _tok(q_nchw).astype(np.float64), _tok(k_nchw).astype(np.float64)
    D = q.shape[2] // nhead

    def fn(b, l, h, idx):
        return (k[b, idx, h * D:(h + 1) * D] @ q[b, l, h * D:(h + 1) * D]) / np.sqrt(D)
    return fn


def test_oracle_on_demo_pair_activations():
    qs, ks, vs = ([F32(f"qta_{n}{lv}") for lv in range(3)] for n in "qkv")
    final, lv = oracle.qtattb_forward(qs, ks, vs, F32("qta_weight"), 8, [32, 16, 8])
    # every series must equal the reference's; differences are accepted only as audited near ties (rank-by-rank equal logits)
    nbad = sum(audit_topk_mismatches(lv[i]["topk_idx"], I64(f"qta_L{i}_topk_idx"), _qta_logit_fn(qs[2 - i], ks[2 - i]),
                                     f"QTAttB level {i} top-k vs reference") for i in range(3))
    if nbad == 0:
        assert_close(final, F32("qta_final"), TOL, "QTAttB final message")
    # cascade attention
    tp = I64("cas_topk_pos")
    q, k, v = (_tok(F32(n)) for n in ("cas_q", "cas_k", "cas_v"))
    msg, up = oracle.cascade_attn(q, k, v, tp, (H4, W4), (H4, W4), 4)
    assert np.array_equal(up[:, ::16], I64("cas_up_idx_sub"))
    assert_close(msg[:, ::4], F32("cas_message_sub"), TOL, "CascadeQTAttB message")
    # coarse matching
    o = oracle.dual_softmax(F32("m8_f0"), F32("m8_f1"), (H8, W8), (H8, W8), 0.1, 0.2, recip=False, want_conf=True)
    audit_index_mismatches(o["next_idx_c01"], I64("m8_next_idx_c01"), dot_score_fn(F32("m8_f0"), F32("m8_f1")), "coarse next_idx_c01")
    audit_index_mismatches(o["next_idx_c10"], I64("m8_next_idx_c10"), dot_score_fn(F32("m8_f1"), F32("m8_f0")), "coarse next_idx_c10")
    assert_close(o["next_conf_c01"], F32("m8_next_conf_c01"), TOL, "coarse next_conf_c01")
    assert_close(o["conf_matrix"].max(2), F32("m8_conf_rowmax"), TOL, "coarse conf row max")
    assert len(match_set(0 * o["i_ids"], o["i_ids"], o["j_ids"]) ^ match_set(0 * I64("m8_i_ids"), I64("m8_i_ids"), I64("m8_j_ids"))) <= 1
    # cascade matching + selection
    d01 = oracle.window_match(F32("m4_f0"), F32("m4_f1"), I64("m4_idx01"), 1.0, recip=False)
    d10 = oracle.window_match(F32("m4_f1"), F32("m4_f0"), I64("m4_idx10"), 1.0, recip=False, want_conf=False)
    audit_index_mismatches(d01["next_idx"], I64("m4_next_idx_c01"), dot_score_fn(F32("m4_f0"), F32("m4_f1")), "cascade next_idx_c01")
    audit_index_mismatches(d10["next_idx"], I64("m4_next_idx_c10"), dot_score_fn(F32("m4_f1"), F32("m4_f0")), "cascade next_idx_c10")
    assert_close(d01["next_conf"], F32("m4_next_conf_c01"), TOL, "cascade next_conf_c01")
    sel = oracle.nms_select(F32("m4_next_conf_c01"), I64("m4_next_idx_c01"), I64("m4_next_idx_c10"), (H4, W4), (H4, W4), 5, 0.2,
                            [(F32("m8_next_conf_c01"), (H8, W8), 0.2)], 2)
    assert np.array_equal(sel["i_ids"], I64("m4_i_ids")) and np.array_equal(sel["j_ids"], I64("m4_j_ids"))


@pytest.mark.gpu
def test_hip_on_demo_pair_activations():
    from casmtr_amd.matching.cascade_matching import CascadeMatching
    from casmtr_amd.matching.coarse_matching import CoarseMatching
    from casmtr_amd.modules.quadtree_attention import CascadeQTAttB, QTAttB
    from casmtr_amd import ops
    dev = "cuda:0"
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    N = lambda t: t.detach().cpu().numpy()
    # QTAttB: module vs reference, per-level indices vs oracle (bit-exact)
    qs, ks, vs = ([F32(f"qta_{n}{lv}") for lv in range(3)] for n in "qkv")
    m = QTAttB(8, 32, scale=3, topks=[32, 16, 8]).to(dev).eval()
    with torch.no_grad():
        m.weight.copy_(T(F32("qta_weight")))
        out = m([T(x) for x in qs], [T(x) for x in ks], [T(x) for x in vs])
    fo, lv = oracle.qtattb_forward(qs, ks, vs, F32("qta_weight"), 8, [32, 16, 8])
    assert_close(N(out), fo, TOL, "QTAttB vs oracle on real-image activations")
    c = ops.qta_coarse_level(T(_tok(qs[2])), T(_tok(ks[2])), T(_tok(vs[2])), 8, 32)
    assert np.array_equal(N(c["topk_idx"]), lv[0]["topk_idx"])
    f1 = ops.qta_fine_level(T(_tok(qs[1])), T(_tok(ks[1])), T(_tok(vs[1])), c["topk_idx"], qs[1].shape[2:], ks[1].shape[2:], 8, 16)
    assert np.array_equal(N(f1["topk_idx"]), lv[1]["topk_idx"])
    if all(np.array_equal(lv[i]["topk_idx"], I64(f"qta_L{i}_topk_idx")) for i in range(3)):
        assert_close(N(out), F32("qta_final"), TOL, "QTAttB vs reference python")
    # CascadeQTAttB
    cm = CascadeQTAttB(4, 32, dilated=1).to(dev)
    with torch.no_grad():
        msg, up = cm(T(F32("cas_q")), T(F32("cas_k")), T(F32("cas_v")), T(I64("cas_topk_pos")), None)
    assert np.array_equal(N(up)[:, ::16], I64("cas_up_idx_sub"))
    assert_close(N(msg)[:, ::4], F32("cas_message_sub"), TOL, "CascadeQTAttB vs reference python")
    # matchers through the data-dict protocol, CPU division convention (the fixture comes from the reference on CPU)
    data = {"hw0_i": (HI, WI), "hw1_i": (HI, WI), "hw0_8c": (H8, W8), "hw1_8c": (H8, W8), "hw0_4c": (H4, W4), "hw1_4c": (H4, W4)}
    c8 = CoarseMatching({"thr": 0.2, "border_rm": 0, "train_coarse_percent": 0.3, "train_pad_num_gt_min": 200,
                         "match_type": "dual_softmax", "dsmax_temperature": 0.1}, div_mode="cpu").eval()
    with torch.no_grad():
        c8(T(F32("m8_f0")), T(F32("m8_f1")), data, level="8c")
    o8 = oracle.dual_softmax(F32("m8_f0"), F32("m8_f1"), (H8, W8), (H8, W8), 0.1, 0.2, recip=False)
    s8 = data["stage_8c"]
    assert np.array_equal(N(s8["next_idx_c01"]), o8["next_idx_c01"]) and np.array_equal(N(s8["next_idx_c10"]), o8["next_idx_c10"])
    assert_close(N(s8["next_conf_c01"]), F32("m8_next_conf_c01"), TOL, "coarse next_conf vs reference python")
    c4 = CascadeMatching({"thr": 0.2, "test_thr": 0.2, "pre_thr": [0.2], "border_rm": 2, "double_check": True,
                          "train_pad_num_gt_min": 200, "match_type": "softmax", "dsmax_temperature": 1.0},
                         {"propagation": "window", "dilated": 1, "post_config": {"method": "maxpool_nms", "window_size": 5}},
                         "4c", div_mode="cpu").eval()
    with torch.no_grad():
        c4(T(F32("m4_f0")), T(F32("m4_f1")), T(I64("m4_idx01")), T(I64("m4_idx10")), data, level="4c", pre_level="8c")
    s4 = data["stage_4c"]
    o01 = oracle.window_match(F32("m4_f0"), F32("m4_f1"), I64("m4_idx01"), 1.0, recip=False)
    assert np.array_equal(N(s4["next_idx_c01"]), o01["next_idx"])
    assert_close(N(s4["next_conf_c01"]), F32("m4_next_conf_c01"), TOL, "cascade next_conf vs reference python")
    assert len(match_set(N(s4["b_ids"]), N(s4["i_ids"]), N(s4["j_ids"])) ^ match_set(0 * I64("m4_i_ids"), I64("m4_i_ids"), I64("m4_j_ids"))) <= 1
