"""CascadeMatching: drop-in for src/model/functions/cascade_matching.py:37-331, inference branch.

forward (:63-168)       -> two casmtr_window_match_fwd launches (0->1 with conf_matrix, 1->0 without)
get_coarse_match (:170) -> one casmtr_nms_select_fwd (maxpool NMS / thresholds / previous-stage confidence / border
                           removal / double check / keep-one fallback / ordered compaction)
The training branch (:264-314, GT window labels, detector heads) is outside the hot path and fails loudly.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .cascade_functions import valid_extents
from .post_processing import PostProcess

INF = 1e9


class CascadeMatching(nn.Module):
    def __init__(self, config, cas_config, stage=None, div_mode="gpu", defer_sync=False, materialize_idx=True):
        super().__init__()
        self.config = config
        self.cas_config = cas_config
        self.thr = config["thr"]
        self.test_thr = config["test_thr"]
        self.pre_thr = config["pre_thr"]
        self.border_rm = config["border_rm"]
        self.double_check = config["double_check"]
        self.train_pad_num_gt_min = config["train_pad_num_gt_min"]
        self.propagation = cas_config["propagation"]
        self.dilated = cas_config["dilated"]
        self.post_process = PostProcess(post_config=cas_config["post_config"])
        self.detector_mode = cas_config.get("detector_mode", None)
        self.grid_size = cas_config.get("grid_size", None)
        self.rt = cas_config["post_config"].get("rt", None)
        self.rd = cas_config["post_config"].get("rd", None)
        if self.rt is not None or self.rd is not None:
            # dead code in the reference: both filters read stage['next_conf_c01_s'] / ['next_idx_c01_s'] (:195,209,217), which
            # CoarseMatching and CascadeMatching.forward always set to None (coarse_matching.py:74, cascade_matching.py:130), so
            # the reference raises TypeError at the first `None / tensor`.  There is no behaviour to reproduce.
            raise NotImplementedError("post_config rt / rd cannot run in the reference either (next_conf_c01_s is always None)")
        self.stage = stage
        self.next_topk = cas_config.get("next_topk", None)
        self.match_type = config["match_type"]
        assert self.match_type == "softmax"
        self.temperature = config["dsmax_temperature"]
        assert div_mode in ("gpu", "cpu")
        self.recip = div_mode == "gpu"
        # defer_sync=True: forward() does not read the match count back (no host sync); the match lists stay capacity-sized
        # until finalize(data, level) is called.  materialize_idx=False: when the window lists arrive as ops.WindowIndex,
        # data['stage_*']['idx_c01'/'idx_c10'] keep that implicit form (call .materialize() for the int64 tensor).
        self.defer_sync = defer_sync
        self.materialize_idx = materialize_idx

    def forward(self, feat_c0, feat_c1, idx_c01, idx_c10, data, mask_c0=None, mask_c1=None, heatmap_c0=None, level="4c",
                pre_level="8c"):
        """idx_c01 / idx_c10: int64 [B,N,K] as in the reference, or ops.WindowIndex (topk_pos + grid sizes)."""
        if self.training:
            raise NotImplementedError(
                "CascadeMatching training branch (GT window labels, cascade_matching.py:264-314) is outside the MI355X hot path: "
                "for training keep the reference's own matcher with casmtr_amd.compat.install(matching=False)")
        hw0, hw1 = tuple(int(x) for x in data[f"hw0_{level}"]), tuple(int(x) for x in data[f"hw1_{level}"])
        f0, f1 = feat_c0.contiguous().float(), feat_c1.contiguous().float()
        implicit = isinstance(idx_c01, ops.WindowIndex)
        if not implicit:
            idx_c01, idx_c10 = idx_c01.contiguous(), idx_c10.contiguous()
        if self.post_process.method == "d2d":
            data["S_d2d"], data["d2d_w"] = self._d2d_scores(f0, hw0), hw0[1] // 4
        d01 = ops.window_match(f0, f1, idx_c01, self.temperature, mask_c0, mask_c1, recip=self.recip, want_conf=True, hw=hw0)
        d10 = ops.window_match(f1, f0, idx_c10, self.temperature, mask_c1, mask_c0, recip=self.recip, want_conf=False, hw=hw1)
        if implicit and self.materialize_idx:
            idx_c01, idx_c10 = idx_c01.materialize(), idx_c10.materialize()
        data[f"stage_{level}"] = {
            "conf_matrix": d01["conf_matrix"], "detector_matrix01": None,
            "next_conf_c01_topk": None, "next_idx_c01_topk": None, "next_conf_c10_topk": None, "next_idx_c10_topk": None,
            "idx_c01": idx_c01, "idx_c10": idx_c10,
            "next_idx_c01": d01["next_idx"], "next_idx_c10": d10["next_idx"],
            "next_conf_c01": d01["next_conf"], "next_conf_c10": d10["next_conf"],
            "next_conf_c01_s": None, "next_idx_c01_s": None,
        }
        match_result = self.get_coarse_match(d01["conf_matrix"], idx_c01, d01["next_conf"], d01["next_idx"], d10["next_idx"],
                                             data, level, pre_level)
        data[f"stage_{level}"].update(**match_result)
        if "m_bids" in match_result:
            data["m_bids"] = match_result["m_bids"]

    @staticmethod
    def _d2d_scores(feat, hw):
        """cascade_matching.py:90-104 ('d2d' detection score on the 1/4 sub-grid): channel spread of the normalised feature at every 4th
        position times the (min-max normalised) norm of its 5 x 5 high-pass response, [B, (h/4)*(w/4), 1].  Plain torch: only the
        unshipped 'd2d' PostProcess reads it."""
        B, N, C = feat.shape
        h, w = hw
        x = (feat / C ** .5).reshape(B, h, w, C)
        spread = x[:, ::4, ::4].std(dim=-1).reshape(B, -1, 1)             # nearest-neighbour x0.25 = every 4th row / column
        xc = x.permute(0, 3, 1, 2)
        box = F.avg_pool2d(xc, 5, stride=4, padding=2, divisor_override=1)   # 5 x 5 sums, zero padded, at the same positions
        ctr = xc[:, :, ::4, ::4]
        resp = ((24.0 + 1.0 / 25.0) * ctr - box / 25.0).norm(dim=1)         # kernel: -1/25 everywhere, 24 at the centre
        resp = (resp - resp.min()) / (resp.max() - resp.min())
        return spread * resp.reshape(B, -1, 1)

    @classmethod
    def finalize(cls, data, level, n=None):
        """Read the match count back (the host sync of `mask.sum() == 0` / torch.where, cascade_matching.py:254-258) and fill
        the list keys.  No-op unless forward ran with defer_sync=True.  `n`: the count if the caller has already read it."""
        st = data[f"stage_{level}"]
        pend = st.pop("_pending", None)
        if pend is None:
            return
        sel, hw0, hw1 = pend
        st.update(**cls._match_dict(sel, int(sel["n"].item()) if n is None else int(n), hw0, hw1, data, level))
        data["m_bids"] = st["m_bids"]

    @staticmethod
    def _match_dict(sel, n, hw0, hw1, data, level):
        b_ids, i_ids, j_ids, mconf = (sel[k][:n] for k in ("b_ids", "i_ids", "j_ids", "mconf"))
        w0, w1 = hw0[1], hw1[1]
        scale = data["hw0_i"][0] / data[f"hw0_{level}"][0]
        scale0 = scale * data["scale0"][b_ids] if "scale0" in data else scale
        scale1 = scale * data["scale1"][b_ids] if "scale1" in data else scale
        mkpts0_c = torch.stack([i_ids % w0, torch.div(i_ids, w0, rounding_mode="trunc")], dim=1) * scale0
        mkpts1_c = torch.stack([j_ids % w1, torch.div(j_ids, w1, rounding_mode="trunc")], dim=1) * scale1
        return {"b_ids": b_ids, "i_ids": i_ids, "j_ids": j_ids, "m_bids": b_ids, "mkpts0_c": mkpts0_c,
                "mkpts1_c": mkpts1_c, "mconf": mconf}

    @torch.no_grad()
    def get_coarse_match(self, conf_matrix01, idx_c01, next_conf_c01, next_idx_c01, next_idx_c10, data, level, pre_level):
        hw0, hw1 = tuple(int(x) for x in data[f"hw0_{level}"]), tuple(int(x) for x in data[f"hw1_{level}"])
        if not isinstance(pre_level, list):
            pre_level = [pre_level]
        pre = []
        for i, pl in enumerate(pre_level):
            pc = data[f"stage_{pl}"]["next_conf_c01"].detach().contiguous().float()
            pre.append((pc, tuple(int(x) for x in data[f"hw0_{pl}"]), float(self.pre_thr[i])))
        valid = None
        if f"mask_{level}0" in data:
            valid = valid_extents(data[f"mask_{level}0"], data[f"mask_{level}1"])
        extra = self.post_process.extra_mask(next_conf_c01, hw0, data, next_idx_c01, hw1)
        sel = ops.nms_select(next_conf_c01, next_idx_c01, next_idx_c10, hw0, hw1, nms_window=self.post_process.nms_window,
                             test_thr=float(self.test_thr), pre=pre, border_rm=int(self.border_rm), valid_hw=valid,
                             double_check=bool(self.double_check), extra_keep=extra)
        if self.defer_sync:
            return {"_pending": (sel, hw0, hw1)}
        # host sync, as `mask.sum() == 0` / torch.where in the reference (:254-258)
        return self._match_dict(sel, int(sel["n"].item()), hw0, hw1, data, level)
