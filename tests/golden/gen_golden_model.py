#!/usr/bin/env python3
"""SURVEY.md §8 f.3 fixture: the reference's whole CasMTR-4c (outdoor stage-3 config, test_single_pair.py's --NMS post-processing)
on the london_bridge demo pair at 256x192, CPU, with the deterministic weights of tests/golden_inputs.model_state().

The model is evaluated stage by stage by calling the reference's own sub-modules; at the two boundaries that feed discrete
decisions (the 1/8 features entering the QuadTree transformer, the 1/8 and 1/4 tokens entering the matchers) the tensors are
rounded to fp16-exact values first, so the fixture stores them compactly AND both implementations see bit-identical inputs there.
Stored: the two resized images (uint8), strided samples of the backbone maps, the rounded stage inputs / outputs, the matchers'
index outputs and the final sub-pixel matches.

    python tests/golden/gen_golden_model.py            (build container only; needs /root/reference)

Nothing of the reference is copied: modules are imported and called, only numbers are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import gen_golden as gg  # noqa: E402  installs the extension / kornia / timm stubs
import ref_stubs  # noqa: E402
import gen_golden_e2e as ge  # noqa: E402
from golden_inputs import model_state  # noqa: E402

REF = gg.REF
THRESHOLDS = dict(coarse_thr=0.002, cascade_thr=0.011, pre_thr=0.0, double_check=False)   # a random-weight network is not confident: test thresholds


def h16(t):
    return t.detach().half().float()


def build_reference(which="4c"):
    ref_stubs.install_full_model_extras()
    from configs.default import get_cfg_defaults
    cfg = get_cfg_defaults()
    stage = "stage3" if which == "4c" else "stage4"
    cfg.merge_from_file(os.path.join(REF, f"configs/model_configs/outdoor/loftr_ds_quadtree_cas_twins_large_{stage}.py"))
    mc = ref_stubs.lower(cfg)["loftr"]
    mc["match_coarse"]["thr"] = THRESHOLDS["coarse_thr"]
    if which == "4c":
        mc["coarse2"]["post_config"]["method"] = "maxpool_nms"
        mc["coarse2"]["post_config"]["window_size"] = 5
        mc["match_cascade"]["test_thr"] = [THRESHOLDS["cascade_thr"]]
        mc["match_cascade"]["pre_thr"] = [[THRESHOLDS["pre_thr"]]]
        mc["match_cascade"]["double_check"] = [THRESHOLDS["double_check"]]
        from src.model.cascade_model_stage3 import CasMTR
    else:
        mc["match_cascade"]["test_thr"] = [THRESHOLDS["cascade_thr"]] * 2
        mc["match_cascade"]["pre_thr"] = [[THRESHOLDS["pre_thr"]], [THRESHOLDS["pre_thr"]] * 2]
        mc["match_cascade"]["double_check"] = [THRESHOLDS["double_check"]] * 2
        from src.model.cascade_model_stage4 import CasMTR
    model = CasMTR(config=mc).eval()
    sd = model.state_dict()
    new = model_state({k: tuple(v.shape) for k, v in sd.items()})
    for k, v in new.items():
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    import json
    with open(os.path.join(HERE, "model_state_keys.json" if which == "4c" else "model2c_state_keys.json"), "w") as f:   # names + shapes of the reference's checkpoint layout
        json.dump({k: list(v.shape) for k, v in sd.items()}, f, indent=0)
    return model


def main():
    model = build_reference()
    im0, im1 = ge.load_pair()
    u8 = [(x * 255.0).round().to(torch.uint8) for x in (im0, im1)]
    im0, im1 = [x.float() / 255.0 for x in u8]
    data = {"image0": im0, "image1": im1, "bs": 1, "hw0_i": im0.shape[2:], "hw1_i": im1.shape[2:]}
    out = {"image0": u8[0], "image1": u8[1]}
    with torch.no_grad():
        f8, f4, ff = model.backbone(torch.cat([im0, im1], 0))
        data.update({"hw0_8c": f8.shape[2:], "hw1_8c": f8.shape[2:], "hw0_4c": f4.shape[2:], "hw1_4c": f4.shape[2:],
                     "hw0_f": ff.shape[2:], "hw1_f": ff.shape[2:]})
        out.update(bb_f8_sub=f8[:, ::4, ::2, ::2].contiguous(), bb_f4_sub=f4[:, ::4, ::4, ::4].contiguous(),
                   bb_ff_sub=ff[:, ::4, ::8, ::8].contiguous())
        # ---- 1/8 stage on fp16-exact features
        f8r = h16(f8)
        out["f8"] = f8r.half()
        t8_0, t8_1 = model.loftr_coarse_8c.forward(model.pos_encoding_8c(f8r[:1]), model.pos_encoding_8c(f8r[1:]), None, None)
        t8_0, t8_1 = h16(t8_0), h16(t8_1)
        out["t8"] = torch.cat([t8_0, t8_1]).half()
        model.coarse_matching_8c.forward(t8_0, t8_1, data, mask_c0=None, mask_c1=None, level="8c")
        s8 = data["stage_8c"]
        out.update(m8_next_idx_c01=s8["next_idx_c01"].to(torch.int16), m8_next_idx_c10=s8["next_idx_c10"].to(torch.int16),
                   m8_next_conf_c01=s8["next_conf_c01"], m8_i_ids=s8["i_ids"].to(torch.int16), m8_j_ids=s8["j_ids"].to(torch.int16),
                   m8_mconf=s8["mconf"])
        # ---- 1/4 stage: the reference's own (unrounded) 1/4 features + the rounded 1/8 tokens
        g = lambda t: t.transpose(1, 2).reshape(1, -1, *f8.shape[2:])
        f4_0, f4_1 = model.up_block1.forward(f4[:1], f4[1:], g(t8_0), g(t8_1), data["hw0_4c"], data["hw1_4c"], 1)
        out["up_sub"] = torch.cat([f4_0, f4_1])[:, ::4, ::4, ::4].contiguous()
        t4_0, t4_1, idx01, idx10, _ = model.loftr_coarse_4c.forward(model.pos_encoding_4c(f4_0), model.pos_encoding_4c(f4_1),
                                                                    s8["next_idx_c01"], s8["next_idx_c10"], data=data)
        t4_0, t4_1 = h16(t4_0), h16(t4_1)
        out["t4"] = torch.cat([t4_0, t4_1]).half()
        model.cascade_matching_4c.forward(t4_0, t4_1, idx01, idx10, data, mask_c0=None, mask_c1=None, heatmap_c0=None, level="4c",
                                          pre_level="8c")
        s4 = data["stage_4c"]
        out.update(m4_b_ids=s4["b_ids"].to(torch.int16), m4_i_ids=s4["i_ids"].to(torch.int16), m4_j_ids=s4["j_ids"].to(torch.int16),
                   m4_mconf=s4["mconf"], m4_mkpts0_c=s4["mkpts0_c"], m4_mkpts1_c=s4["mkpts1_c"])
        # ---- fine stage on the reference's match list
        w0, w1 = model.fine_preprocess.forward(ff[:1], ff[1:], t4_0, t4_1, data)
        if w0.size(0):
            w0, w1 = model.loftr_fine(w0, w1)
        model.fine_matching.forward(w0.float(), w1.float(), data)
        out.update(mkpts0_f=data["mkpts0_f"], mkpts1_f=data["mkpts1_f"], expec_f=data["expec_f"])
    out["thresholds"] = np.array([THRESHOLDS["coarse_thr"], THRESHOLDS["cascade_thr"], THRESHOLDS["pre_thr"], float(THRESHOLDS["double_check"])],
                                 dtype=np.float32)
    arrs = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()}
    path = os.path.join(HERE, "model_london_bridge.npz")
    np.savez_compressed(path, **arrs)
    print(f"model_london_bridge: {os.path.getsize(path) / 1e6:.2f} MB; coarse matches {len(arrs['m8_i_ids'])}, "
          f"cascade matches {len(arrs['m4_i_ids'])}; |t8| {float(t8_0.abs().mean()):.3f} |t4| {float(t4_0.abs().mean()):.3f} "
          f"conf8 max {float(s8['next_conf_c01'].max()):.3f} mconf4 mean {float(s4['mconf'].mean()) if len(s4['mconf']) else -1:.3f} "
          f"expec std {float(data['expec_f'][:, :2].std()) if len(data['expec_f']) else -1:.3f}")


def main_2c():
    """CasMTR-2c (cascade_model_stage4.py): what the third stage adds -- up_block2, loftr_coarse_2c, cascade_matching_2c, the fine
    refinement on the 1/2-level tokens -- evaluated on fp16-exact 1/4 tokens.  256x128 keeps the 1/2-level token fixture at 2 MB."""
    model = build_reference("2c")
    im0, im1 = ge.load_pair(hw=(128, 256))
    u8 = [(x * 255.0).round().to(torch.uint8) for x in (im0, im1)]
    im0, im1 = [x.float() / 255.0 for x in u8]
    data = {"image0": im0, "image1": im1, "bs": 1, "hw0_i": im0.shape[2:], "hw1_i": im1.shape[2:]}
    out = {"image0": u8[0], "image1": u8[1]}
    with torch.no_grad():
        f8, f4, ff = model.backbone(torch.cat([im0, im1], 0))
        for lv, f in (("8c", f8), ("4c", f4), ("2c", ff), ("f", ff)):
            data[f"hw0_{lv}"] = data[f"hw1_{lv}"] = f.shape[2:]
        t8_0, t8_1 = model.loftr_coarse_8c.forward(model.pos_encoding_8c(f8[:1]), model.pos_encoding_8c(f8[1:]), None, None)
        model.coarse_matching_8c.forward(t8_0.float(), t8_1.float(), data, mask_c0=None, mask_c1=None, level="8c")
        s8 = data["stage_8c"]
        g = lambda t, f: t.transpose(1, 2).reshape(1, -1, *f.shape[2:])
        f4_0, f4_1 = model.up_block1.forward(f4[:1], f4[1:], g(t8_0, f8), g(t8_1, f8), data["hw0_4c"], data["hw1_4c"], 1)
        t4_0, t4_1, i01, i10, _ = model.loftr_coarse_4c.forward(model.pos_encoding_4c(f4_0), model.pos_encoding_4c(f4_1),
                                                                s8["next_idx_c01"], s8["next_idx_c10"], data=data)
        t4_0, t4_1 = h16(t4_0), h16(t4_1)
        out["t4"] = torch.cat([t4_0, t4_1]).half()
        model.cascade_matching_4c.forward(t4_0, t4_1, i01, i10, data, mask_c0=None, mask_c1=None, heatmap_c0=None, level="4c", pre_level="8c")
        s4 = data["stage_4c"]
        out.update(m8_next_conf_c01=s8["next_conf_c01"], m4_next_conf_c01=s4["next_conf_c01"],
                   m4_next_idx_c01=s4["next_idx_c01"].to(torch.int16), m4_next_idx_c10=s4["next_idx_c10"].to(torch.int16))
        # ---- the third stage on the reference's own 1/2 backbone map + the rounded 1/4 tokens
        f2_0, f2_1 = model.up_block2.forward(ff[:1], ff[1:], g(t4_0, f4), g(t4_1, f4), data["hw0_2c"], data["hw1_2c"], 1)
        out["up2_sub"] = torch.cat([f2_0, f2_1])[:, ::4, ::4, ::4].contiguous()
        t2_0, t2_1, j01, j10, _ = model.loftr_coarse_2c.forward(model.pos_encoding_2c(f2_0), model.pos_encoding_2c(f2_1),
                                                                s4["next_idx_c01"], s4["next_idx_c10"], data=data)
        t2_0, t2_1 = h16(t2_0), h16(t2_1)
        out["t2"] = torch.cat([t2_0, t2_1]).half()
        model.cascade_matching_2c.forward(t2_0, t2_1, j01, j10, data, mask_c0=None, mask_c1=None, heatmap_c0=None, level="2c",
                                          pre_level=["8c", "4c"])
        s2 = data["stage_2c"]
        out.update(m2_b_ids=s2["b_ids"].to(torch.int16), m2_i_ids=s2["i_ids"].to(torch.int32), m2_j_ids=s2["j_ids"].to(torch.int32),
                   m2_mconf=s2["mconf"], m2_mkpts0_c=s2["mkpts0_c"], m2_mkpts1_c=s2["mkpts1_c"])
        w0, w1 = model.fine_preprocess.forward(g(t2_0, ff), g(t2_1, ff), feat_c0=None, feat_c1=None, data=data)
        if w0.size(0):
            w0, w1 = model.loftr_fine(w0, w1)
        model.fine_matching.forward(w0.float(), w1.float(), data)
        out.update(mkpts0_f=data["mkpts0_f"], mkpts1_f=data["mkpts1_f"], expec_f=data["expec_f"])
    out["thresholds"] = np.array([THRESHOLDS["coarse_thr"], THRESHOLDS["cascade_thr"], THRESHOLDS["pre_thr"], float(THRESHOLDS["double_check"])],
                                 dtype=np.float32)
    arrs = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()}
    path = os.path.join(HERE, "model2c_london_bridge.npz")
    np.savez_compressed(path, **arrs)
    print(f"model2c_london_bridge: {os.path.getsize(path) / 1e6:.2f} MB; 1/4 matches {len(s4['i_ids'])}, 1/2 matches {len(arrs['m2_i_ids'])}; "
          f"|t2| {float(t2_0.abs().mean()):.3f}; expec std {float(data['expec_f'][:, :2].std()) if len(data['expec_f']) else -1:.3f}")


def main_indoor():
    """the indoor model (cascade_quadtree_stage3.py, indoor stage-3 config): ResNet-FPN + ladder, 8 QuadTree layers with top-k
    [32,16,16], POLA self-attention, cascade cross-attention with the learned relative position bias, no NMS.  256x128."""
    ref_stubs.install_full_model_extras()
    from configs.default import get_cfg_defaults
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(REF, "configs/model_configs/indoor/loftr_ds_quadtree_cas_stage3.py"))
    mc = ref_stubs.lower(cfg)["loftr"]
    mc["match_coarse"]["thr"] = THRESHOLDS["coarse_thr"]
    mc["match_cascade"]["test_thr"] = [THRESHOLDS["cascade_thr"]]
    mc["match_cascade"]["pre_thr"] = [[THRESHOLDS["pre_thr"]] * 2]
    mc["match_cascade"]["double_check"] = [THRESHOLDS["double_check"]]
    from src.model.cascade_quadtree_stage3 import CasMTR
    model = CasMTR(config=mc).eval()
    sd = model.state_dict()
    for k, v in model_state({k: tuple(v.shape) for k, v in sd.items()}).items():
        if sd[k].dtype == torch.float32:      # buffers of integer type (relative_position_index) keep their constructor values
            sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    import json
    with open(os.path.join(HERE, "model_indoor_state_keys.json"), "w") as f:
        json.dump({k: list(v.shape) for k, v in sd.items()}, f, indent=0)
    im0, im1 = ge.load_pair(hw=(128, 256))
    u8 = [(x * 255.0).round().to(torch.uint8) for x in (im0, im1)]
    im0, im1 = [x.float() / 255.0 for x in u8]
    x = torch.cat([im0, im1], 0)
    data = {"image0": im0, "image1": im1, "bs": 1, "hw0_i": im0.shape[2:], "hw1_i": im1.shape[2:]}
    out = {"image0": u8[0], "image1": u8[1]}
    with torch.no_grad():
        f8, f4, ff = model.backbone(x)
        for lv, f in (("c", f8), ("8c", f8), ("4c", f4), ("2c", ff), ("f", ff)):
            data[f"hw0_{lv}"] = data[f"hw1_{lv}"] = f.shape[2:]
        out.update(bb_f8_sub=f8[:, ::4, ::2, ::2].contiguous(), bb_f4_sub=f4[:, ::4, ::2, ::2].contiguous(), bb_ff_sub=ff[:, ::4, ::4, ::4].contiguous())
        f8r = h16(f8)
        out["f8"] = f8r.half()
        t8_0, t8_1 = model.loftr_coarse.forward(model.pos_encoding(f8r[:1]), model.pos_encoding(f8r[1:]), None, None)
        t8_0, t8_1 = h16(t8_0), h16(t8_1)
        out["t8"] = torch.cat([t8_0, t8_1]).half()
        model.coarse_matching.forward(t8_0, t8_1, data, mask_c0=None, mask_c1=None, level="8c")
        s8 = data["stage_8c"]
        out.update(m8_next_idx_c01=s8["next_idx_c01"].to(torch.int16), m8_next_idx_c10=s8["next_idx_c10"].to(torch.int16),
                   m8_next_conf_c01=s8["next_conf_c01"])
        l4, lf = model.ladder.forward(x, [f4, ff])
        out.update(lad_4_sub=l4[:, ::4, ::2, ::2].contiguous(), lad_f_sub=lf[:, ::4, ::4, ::4].contiguous())
        g = lambda t, f: t.transpose(1, 2).reshape(1, -1, *f.shape[2:])
        f4_0, f4_1 = model.up_block1.forward(l4[:1], l4[1:], g(t8_0, f8), g(t8_1, f8), data["hw0_4c"], data["hw1_4c"], 1)
        t4_0, t4_1, i01, i10, _ = model.loftr_coarse_4c.forward(model.pos_encoding_4c(f4_0), model.pos_encoding_4c(f4_1),
                                                                s8["next_idx_c01"], s8["next_idx_c10"], data=data)
        t4_0, t4_1 = h16(t4_0), h16(t4_1)
        out["t4"] = torch.cat([t4_0, t4_1]).half()
        model.cascade_matching_4c.forward(t4_0, t4_1, i01, i10, data, mask_c0=None, mask_c1=None, heatmap_c0=None, level="4c", pre_level="8c")
        s4 = data["stage_4c"]
        out.update(m4_b_ids=s4["b_ids"].to(torch.int16), m4_i_ids=s4["i_ids"].to(torch.int16), m4_j_ids=s4["j_ids"].to(torch.int16),
                   m4_mconf=s4["mconf"], m4_mkpts0_c=s4["mkpts0_c"], m4_mkpts1_c=s4["mkpts1_c"])
        w0, w1 = model.cas_fine_preprocess.forward(lf[:1], lf[1:], t4_0, t4_1, data=data)
        if w0.size(0):
            w0, w1 = model.cas_loftr_fine(w0, w1)
        model.cas_fine_matching.forward(w0.float(), w1.float(), data)
        out.update(mkpts0_f=data["mkpts0_f"], mkpts1_f=data["mkpts1_f"], expec_f=data["expec_f"])
    out["thresholds"] = np.array([THRESHOLDS["coarse_thr"], THRESHOLDS["cascade_thr"], THRESHOLDS["pre_thr"], float(THRESHOLDS["double_check"])],
                                 dtype=np.float32)
    arrs = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()}
    path = os.path.join(HERE, "model_indoor_london_bridge.npz")
    np.savez_compressed(path, **arrs)
    print(f"model_indoor_london_bridge: {os.path.getsize(path) / 1e6:.2f} MB; coarse matches {len(s8['i_ids'])}, 1/4 matches {len(arrs['m4_i_ids'])}; "
          f"|t8| {float(t8_0.abs().mean()):.3f} |t4| {float(t4_0.abs().mean()):.3f}; expec std {float(data['expec_f'][:, :2].std()) if len(data['expec_f']) else -1:.3f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "indoor":
        main_indoor()
    elif len(sys.argv) > 1 and sys.argv[1] == "2c":
        main_2c()
    else:
        main()
