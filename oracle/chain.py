"""The hot-path chain of casmtr_amd/pipeline.py::HotPath restated over the CPU oracle, for ONE pair of a synthetic batch.

Test infrastructure (like everything under oracle/): used by tests/ (chain-level parity for every BASELINE config) and by bench.py's
cpu_baseline leg (timed there).  Follows the same call order as the reference model's forward
(src/model/cascade_model_stage3.py:104-178, cascade_model_stage4.py:150-195):
  2L x QTAttB.forward -> CoarseMatching -> per cascade stage: window indices from the previous stage's argmax
  (transformer.py:416-440), 2c x CascadeQTAttB.forward, CascadeMatching both directions, NMS / thresholds / borders / double check.
"""
import numpy as np

from . import (cascade_attn, dual_softmax, nms_select, qtattb_forward, window_match, window_warp_idx)


def run_chain(cfg, inp, pair=0, qta_calls=None):
    """cfg: casmtr_amd.pipeline.HotPathConfig; inp: make_synthetic_inputs(...) (torch tensors, any device); pair: batch index.
    qta_calls: indices into the 2*coarse_layers QTAttB calls to run (None = all; tests at the full size pick a few).
    -> dict(qta={call: (final, levels)}, d8=CoarseMatching outputs, stages={level: dict(msgs, m01, m10, sel, tp01, tp10, i01, i10)})"""
    n = lambda t: t[pair:pair + 1].detach().cpu().numpy()
    li = (lambda layer: layer) if cfg.fresh_inputs else (lambda layer: 0)
    w = inp["weight"].detach().cpu().numpy()
    qta = {}
    call = 0
    for layer in range(cfg.coarse_layers):
        pairs = ((0, 0), (1, 1)) if layer % 2 == 0 else ((0, 1), (1, 0))   # 'self' / 'cross' (transformer.py:294-303)
        for a, b in pairs:
            if qta_calls is None or call in qta_calls:
                qta[call] = qtattb_forward([n(x) for x in inp[f"cq{a}"][li(layer)]], [n(x) for x in inp[f"ck{b}"][li(layer)]],
                                           [n(x) for x in inp[f"cv{b}"][li(layer)]], w, cfg.coarse_heads, cfg.coarse_topks)
            call += 1
    mk = lambda lvl, im: n(inp[f"mask_{lvl}{im}"]).reshape(1, -1) if cfg.masked else None

    def vh(lvl):   # per-pair valid extents (h0, w0, h1, w1), cascade_functions.py:108-109
        if not cfg.masked:
            return None
        m0, m1 = n(inp[f"mask_{lvl}0"]), n(inp[f"mask_{lvl}1"])
        return np.stack([m0.sum(1).max(-1), m0.sum(2).max(-1), m1.sum(1).max(-1), m1.sum(2).max(-1)], 1).astype(np.int32)

    d8 = dual_softmax(n(inp["feat_8c0"]), n(inp["feat_8c1"]), cfg.hw8, cfg.hw8, cfg.coarse_temperature, cfg.coarse_thr,
                      cfg.coarse_border_rm, mask0=mk("8c", 0), mask1=mk("8c", 1), valid_hw=vh("8c"), recip=True)
    stages, prev, pre = {}, d8, [(d8["next_conf_c01"], cfg.hw8)]
    for st in cfg.stages:
        lvl, (h, w_) = st.level, cfg.hw(st.div)
        tp01 = window_warp_idx(prev["next_idx_c01"], h // 2, w_ // 2, cfg.window_size)
        tp10 = window_warp_idx(prev["next_idx_c10"], h // 2, w_ // 2, cfg.window_size)
        tok = lambda x: np.ascontiguousarray(n(x).transpose(0, 2, 3, 1).reshape(1, -1, st.dim))
        rel = lambda k: n(inp[f"{lvl}rel{k}"]) if st.rel_pos else None
        msgs = []
        i01 = i10 = None
        for layer in range(st.cross_layers):
            m0, i01 = cascade_attn(tok(inp[f"{lvl}q0"][li(layer)]), tok(inp[f"{lvl}k1"][li(layer)]), tok(inp[f"{lvl}v1"][li(layer)]),
                                   tp01, (h, w_), (h, w_), st.heads, rel_pos=rel("01"))
            m1, i10 = cascade_attn(tok(inp[f"{lvl}q1"][li(layer)]), tok(inp[f"{lvl}k0"][li(layer)]), tok(inp[f"{lvl}v0"][li(layer)]),
                                   tp10, (h, w_), (h, w_), st.heads, rel_pos=rel("10"))
            msgs += [m0, m1]
        m01 = window_match(n(inp[f"feat_{lvl}0"]), n(inp[f"feat_{lvl}1"]), i01, st.temperature, mk(lvl, 0), mk(lvl, 1), recip=True)
        m10 = window_match(n(inp[f"feat_{lvl}1"]), n(inp[f"feat_{lvl}0"]), i10, st.temperature, mk(lvl, 1), mk(lvl, 0), recip=True,
                           want_conf=False)
        sel = nms_select(m01["next_conf"], m01["next_idx"], m10["next_idx"], (h, w_), (h, w_), st.nms_window, st.test_thr,
                         [(pc, phw, thr) for (pc, phw), thr in zip(pre, st.pre_thr)], st.border_rm, valid_hw=vh(lvl))
        stages[lvl] = dict(msgs=msgs, m01=m01, m10=m10, sel=sel, tp01=tp01, tp10=tp10, i01=i01, i10=i10)
        prev = dict(next_idx_c01=m01["next_idx"], next_idx_c10=m10["next_idx"])
        pre.append((m01["next_conf"], (h, w_)))
    return dict(qta=qta, d8=d8, stages=stages)
