"""CoarseMatching: drop-in for src/model/functions/coarse_matching.py:22-153 (dual-softmax correlation volume, row /
column argmax, thresholded mutual-nearest matches), computed by casmtr_dual_softmax_fwd.

Differences a user can see, all opt-in:
  * materialize_conf=False leaves data['stage_*']['conf_matrix'] = None (nothing downstream of the matcher reads it
    at inference; it costs 4*L*S bytes per pair);
  * div_mode: 'gpu' (default) scales by fl32(1/s) like torch's GPU kernels, 'cpu' divides like torch on CPU;
  * gemm='split': the similarity matrix on the f16 matrix pipe (2.4x faster; ops.ds_gemm_mode explains what stays exact -- the
    argmax indices -- and what carries the 1e-4 softmax tolerance -- confidences and hence borderline match-list entries).  The
    default ('exact') computes every logit with the oracle's fp32 chain.
"""
import torch
import torch.nn as nn

from .. import ops
from .cascade_functions import valid_extents

INF = 1e9


class CoarseMatching(nn.Module):
    def __init__(self, config, coarse_config=None, materialize_conf=True, div_mode="gpu", defer_sync=False, gemm=None):
        super().__init__()
        self.config = config
        self.thr = config["thr"]
        self.border_rm = config["border_rm"]
        self.train_coarse_percent = config["train_coarse_percent"]
        self.train_pad_num_gt_min = config["train_pad_num_gt_min"]
        self.next_topk = coarse_config.get("next_topk", None) if coarse_config is not None else None
        self.match_type = config["match_type"]
        self.temperature = config["dsmax_temperature"]
        self.materialize_conf = materialize_conf
        assert div_mode in ("gpu", "cpu")
        self.recip = div_mode == "gpu"
        # defer_sync=True: forward() does not read the match count back (no host sync); the match lists stay
        # capacity-sized until finalize(data, level) is called.  Nothing on the cascade path needs them earlier.
        self.defer_sync = defer_sync
        assert gemm in (None, "exact", "split")
        self.gemm = gemm

    def _forward_autograd(self, feat_c0, feat_c1, mask_c0, mask_c1):
        """differentiable formulation for training (torch ops on the GPU), coarse_matching.py:62-71"""
        C = feat_c0.shape[-1]
        f0, f1 = feat_c0 / C ** 0.5, feat_c1 / C ** 0.5
        sim = torch.einsum("nlc,nsc->nls", f0, f1) / self.temperature
        if mask_c0 is not None:
            sim = sim.masked_fill(~(mask_c0[..., None] * mask_c1[:, None]).bool(), -INF)
        s10, s01 = torch.softmax(sim, 1), torch.softmax(sim, 2)
        c01, i01 = torch.max(s01, dim=2)
        c10, i10 = torch.max(s10, dim=1)
        return s10 * s01, i01, c01, i10, c10

    def forward(self, feat_c0, feat_c1, data, mask_c0=None, mask_c1=None, level="8c"):
        assert self.match_type == "dual_softmax"
        hw0, hw1 = tuple(int(x) for x in data[f"hw0_{level}"]), tuple(int(x) for x in data[f"hw1_{level}"])
        valid = None
        if f"mask_{level}0" in data:
            valid = valid_extents(data[f"mask_{level}0"], data[f"mask_{level}1"])
        if torch.is_grad_enabled() and (feat_c0.requires_grad or feat_c1.requires_grad):
            conf, i01, c01, i10, c10 = self._forward_autograd(feat_c0, feat_c1, mask_c0, mask_c1)
            with torch.no_grad():
                out = ops.dual_softmax(feat_c0.detach().contiguous().float(), feat_c1.detach().contiguous().float(), hw0,
                                       hw1, self.temperature, self.thr, self.border_rm, mask_c0, mask_c1, valid,
                                       recip=self.recip, want_conf=False, gemm=self.gemm)
            out.update(conf_matrix=conf, next_idx_c01=i01, next_conf_c01=c01, next_idx_c10=i10, next_conf_c10=c10)
        else:
            out = ops.dual_softmax(feat_c0.contiguous().float(), feat_c1.contiguous().float(), hw0, hw1, self.temperature,
                                   self.thr, self.border_rm, mask_c0, mask_c1, valid, recip=self.recip,
                                   want_conf=self.materialize_conf, gemm=self.gemm)
        data[f"stage_{level}"] = {
            "conf_matrix": out["conf_matrix"],
            "next_conf_c01_topk": None, "next_idx_c01_topk": None, "next_conf_c10_topk": None, "next_idx_c10_topk": None,
            "next_idx_c01": out["next_idx_c01"], "next_idx_c10": out["next_idx_c10"],
            "next_conf_c01": out["next_conf_c01"], "next_conf_c10": out["next_conf_c10"],
            "next_conf_c01_s": None, "next_idx_c01_s": None,
        }
        data[f"stage_{level}"]["_pending"] = out
        if not self.defer_sync:
            self.finalize(data, level)

    @classmethod
    def finalize(cls, data, level="8c", n=None):
        """Read the match count back (one host sync, as torch.where at coarse_matching.py:126) and fill the list keys.
        `n`: the count if the caller has already read it (HotPath.finalize reads every stage's count in one transfer)."""
        st = data[f"stage_{level}"]
        out = st.pop("_pending", None)
        if out is None:
            return
        n = int(out["n"].item()) if n is None else int(n)
        b_ids, i_ids, j_ids, mconf = (out[k][:n] for k in ("b_ids", "i_ids", "j_ids", "mconf"))
        st.update(**cls._match_dict(b_ids, i_ids, j_ids, mconf, data, level))

    @staticmethod
    def _match_dict(b_ids, i_ids, j_ids, mconf, data, level):
        """coarse_matching.py:134-151"""
        w0, w1 = int(data[f"hw0_{level}"][1]), int(data[f"hw1_{level}"][1])
        scale = data["hw0_i"][0] / data[f"hw0_{level}"][0]
        scale0 = scale * data["scale0"][b_ids] if "scale0" in data else scale
        scale1 = scale * data["scale1"][b_ids] if "scale1" in data else scale
        mkpts0_c = torch.stack([i_ids % w0, torch.div(i_ids, w0, rounding_mode="trunc")], dim=1) * scale0
        mkpts1_c = torch.stack([j_ids % w1, torch.div(j_ids, w1, rounding_mode="trunc")], dim=1) * scale1
        keep = torch.nonzero(mconf != 0).squeeze(1)   # one host sync for the four filtered lists (boolean indexing syncs per list)
        return {"b_ids": b_ids, "i_ids": i_ids, "j_ids": j_ids, "gt_mask": mconf == 0, "m_bids": b_ids[keep],
                "mkpts0_c": mkpts0_c[keep], "mkpts1_c": mkpts1_c[keep], "mconf": mconf[keep]}
