#!/usr/bin/env python3
"""BASELINE configs[0] as a parity case: the reference's whole CasMTR-4c model (random init -- no checkpoints here) is run
on the london_bridge demo pair on CPU, the tensors that ENTER its hot-path modules are captured, rounded to fp16-exact
values (so the fixture can hold them compactly and bit-exactly), and the reference's hot-path modules are run again on
exactly those tensors.  Stored: the rounded inputs (fp16) and the reference's outputs.  Real-image activations have the
spatial coherence (and the flat regions) that seeded noise lacks.

    python tests/golden/gen_golden_e2e.py            (build container only; needs /root/reference)

Image size 256x192 instead of the demo's 640x480 keeps the fixture at a few MB (coarse grid 32x24, cascade 64x48).
Nothing of the reference is copied: modules are imported and called, only numbers are stored.
"""
import copy
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  installs the extension / kornia / timm stubs and imports the hot-path modules

REF = gg.REF


import ref_stubs  # noqa: E402

CN, lower, more_stubs = ref_stubs.CN, ref_stubs.lower, ref_stubs.install_full_model_extras


def load_pair(hw=(192, 256)):
    from PIL import Image
    d = os.path.join(REF, "assets", "demo_imgs")
    out = []
    for f in ("london_bridge_19481797_2295892421.jpg", "london_bridge_49190386_5209386933.jpg"):
        im = Image.open(os.path.join(d, f)).convert("RGB").resize((hw[1], hw[0]), Image.BILINEAR)
        out.append(torch.from_numpy(np.asarray(im)).permute(2, 0, 1).float()[None] / 255.0)
    return out


def h16(t):
    return t.detach().half().float()


def main():
    more_stubs()
    from configs.default import get_cfg_defaults
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(REF, "configs/model_configs/outdoor/loftr_ds_quadtree_cas_twins_large_stage3.py"))
    mc = lower(cfg)["loftr"]
    mc["coarse2"]["post_config"]["method"] = "maxpool_nms"      # test_single_pair.py --NMS
    mc["coarse2"]["post_config"]["window_size"] = 5
    from src.model.cascade_model_stage3 import CasMTR
    torch.manual_seed(0)
    model = CasMTR(config=mc).eval()

    cap = {}
    qta, CoarseMatching, CascadeMatching = gg.qta, gg.CoarseMatching, gg.CascadeMatching
    orig = {"q": qta.QTAttB.forward, "c": qta.CascadeQTAttB.forward, "m8": CoarseMatching.forward, "m4": CascadeMatching.forward}
    count = {"q": 0, "c": 0}

    def fq(self, queries, keys, values, *a, **k):
        if count["q"] == 2:   # first 'cross' call: image 0 attends to image 1
            cap["q"] = dict(q=[h16(x) for x in queries], k=[h16(x) for x in keys], v=[h16(x) for x in values],
                            weight=self.weight.detach().clone(), nhead=self.nhead, topks=list(self.topks))
        count["q"] += 1
        return orig["q"](self, queries, keys, values, *a, **k)

    def fc(self, query, key, value, topk_pos, rel_pos):
        if count["c"] == 0:
            cap["c"] = dict(q=h16(query), k=h16(key), v=h16(value), topk_pos=topk_pos.clone(), nhead=self.nhead)
        count["c"] += 1
        return orig["c"](self, query, key, value, topk_pos, rel_pos)

    def fm8(self, feat_c0, feat_c1, data, **k):
        cap["m8"] = dict(f0=h16(feat_c0), f1=h16(feat_c1), hw0=tuple(data["hw0_8c"]), hw1=tuple(data["hw1_8c"]), hwi=tuple(data["hw0_i"]))
        return orig["m8"](self, feat_c0, feat_c1, data, **k)

    def fm4(self, feat_c0, feat_c1, idx_c01, idx_c10, data, **k):
        cap["m4"] = dict(f0=h16(feat_c0), f1=h16(feat_c1), idx01=idx_c01.clone(), idx10=idx_c10.clone(),
                         hw0=tuple(data["hw0_4c"]), hw1=tuple(data["hw1_4c"]))
        return orig["m4"](self, feat_c0, feat_c1, idx_c01, idx_c10, data, **k)

    qta.QTAttB.forward, qta.CascadeQTAttB.forward = fq, fc
    CoarseMatching.forward, CascadeMatching.forward = fm8, fm4
    im0, im1 = load_pair()
    with torch.no_grad():
        model({"image0": im0, "image1": im1})
    qta.QTAttB.forward, qta.CascadeQTAttB.forward = orig["q"], orig["c"]
    CoarseMatching.forward, CascadeMatching.forward = orig["m8"], orig["m4"]

    out = {}
    # ---- QTAttB on the captured (fp16-exact) pyramids
    c = cap["q"]
    m = qta.QTAttB(c["nhead"], c["q"][0].shape[1] // c["nhead"], scale=3, topks=c["topks"])
    with torch.no_grad():
        m.weight.copy_(c["weight"])
    rec = gg._record_levels(m)
    with torch.no_grad():
        final = m(c["q"], c["k"], c["v"])
    for lv in range(3):
        for n in "qkv":
            out[f"qta_{n}{lv}"] = c[n][lv].half()
        out[f"qta_L{lv}_topk_idx"] = rec[lv][3].to(torch.int16)
    out["qta_weight"], out["qta_final"] = c["weight"], final
    # ---- CascadeQTAttB
    c = cap["c"]
    mcas = qta.CascadeQTAttB(c["nhead"], c["q"].shape[1] // c["nhead"], dilated=1)
    with torch.no_grad():
        msg, up = mcas(c["q"], c["k"], c["v"], c["topk_pos"], None)
    out.update(cas_q=c["q"].half(), cas_k=c["k"].half(), cas_v=c["v"].half(), cas_topk_pos=c["topk_pos"].to(torch.int16),
               cas_message_sub=msg[:, ::4].contiguous(), cas_up_idx_sub=up[:, ::16].to(torch.int16))
    # ---- CoarseMatching
    c = cap["m8"]
    cm = CoarseMatching(gg.match_config({})).eval()
    data = {"hw0_i": c["hwi"], "hw1_i": c["hwi"], "hw0_8c": c["hw0"], "hw1_8c": c["hw1"]}
    with torch.no_grad():
        cm.forward(c["f0"], c["f1"], data, level="8c")
    s8 = data["stage_8c"]
    out.update(m8_f0=c["f0"].half(), m8_f1=c["f1"].half(), m8_next_idx_c01=s8["next_idx_c01"].to(torch.int16),
               m8_next_idx_c10=s8["next_idx_c10"].to(torch.int16), m8_next_conf_c01=s8["next_conf_c01"],
               m8_i_ids=s8["i_ids"].to(torch.int16), m8_j_ids=s8["j_ids"].to(torch.int16), m8_mconf=s8["mconf"],
               m8_conf_rowmax=s8["conf_matrix"].max(2)[0])
    # ---- CascadeMatching (NMS on, border_rm 2, double check)
    c = cap["m4"]
    mcfg = {"thr": 0.2, "test_thr": 0.2, "pre_thr": [0.2], "border_rm": 2, "double_check": True, "train_pad_num_gt_min": 200,
            "match_type": "softmax", "dsmax_temperature": 1.0}
    cmod = CascadeMatching(mcfg, {"propagation": "window", "dilated": 1, "post_config": {"method": "maxpool_nms", "window_size": 5}},
                           stage="4c").eval()
    data.update({"hw0_4c": c["hw0"], "hw1_4c": c["hw1"]})
    with torch.no_grad():
        cmod.forward(c["f0"], c["f1"], c["idx01"], c["idx10"], data, level="4c", pre_level="8c")
    s4 = data["stage_4c"]
    out.update(m4_f0=c["f0"].half(), m4_f1=c["f1"].half(), m4_idx01=c["idx01"].to(torch.int16), m4_idx10=c["idx10"].to(torch.int16),
               m4_next_idx_c01=s4["next_idx_c01"].to(torch.int16), m4_next_idx_c10=s4["next_idx_c10"].to(torch.int16),
               m4_next_conf_c01=s4["next_conf_c01"], m4_i_ids=s4["i_ids"].to(torch.int16), m4_j_ids=s4["j_ids"].to(torch.int16),
               m4_mconf=s4["mconf"])
    out["meta"] = np.array([c["hw0"][0], c["hw0"][1], cap["m8"]["hw0"][0], cap["m8"]["hw0"][1], cap["m8"]["hwi"][0], cap["m8"]["hwi"][1]])
    arrs = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()}
    path = os.path.join(HERE, "e2e_london_bridge.npz")
    np.savez_compressed(path, **arrs)
    print(f"e2e_london_bridge: {os.path.getsize(path) / 1e6:.2f} MB; coarse matches {len(arrs['m8_i_ids'])}, cascade matches {len(arrs['m4_i_ids'])}")


if __name__ == "__main__":
    main()
