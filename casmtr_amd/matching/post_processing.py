"""Counterpart of src/model/functions/post_processing.py.

`None` (:42-43) and 'maxpool_nms' (:111-121) -- the methods every shipped config selects -- are folded into
casmtr_nms_select_fwd.  'local_window_nms' (:76-93, top-k per non-overlapping window) is built from torch ops on the GPU and
handed to the same kernel as an extra keep mask.  'sift', 'softargmax_nms' and 'd2d' are kornia pipelines (ScaleSpaceDetector,
ConvSoftArgmax2d) or need a detector head's `S_d2d`; kornia is not available in this image, their results could not be pinned
against the reference, and no shipped config selects them: they fail loudly."""
import torch

from .. import ops


class PostProcess(object):
    def __init__(self, post_config):
        self.config = post_config
        self.method = post_config["method"]
        if self.method not in (None, "maxpool_nms", "local_window_nms"):
            raise NotImplementedError(f"PostProcess method {self.method!r} needs kornia and is outside the MI355X hot path")
        if self.method == "maxpool_nms" and post_config.get("stride", 1) != 1:
            raise NotImplementedError("maxpool_nms is implemented for stride 1 (every shipped config)")

    @property
    def nms_window(self):
        return int(self.config["window_size"]) if self.method == "maxpool_nms" else 0

    def extra_mask(self, next_conf_c01, hw0):
        """-> bool [B, H0*W0] for the methods that are not folded into the selection kernel, else None."""
        if self.method != "local_window_nms":
            return None
        B = next_conf_c01.shape[0]
        h, w = hw0
        ws, topk = int(self.config["window_size"]), int(self.config["topk"])
        t = next_conf_c01.reshape(B, h // ws, ws, w // ws, ws).permute(0, 1, 3, 2, 4).reshape(B, -1, ws * ws)
        idx = torch.topk(t, k=topk, dim=2)[1]
        keep = torch.zeros_like(t, dtype=torch.bool).scatter_(2, idx, True)
        return keep.reshape(B, h // ws, w // ws, ws, ws).permute(0, 1, 3, 2, 4).reshape(B, h * w).contiguous()

    def apply(self, data, axes_lengths, next_idx_c01, next_conf_c01, test_thr, level):
        """-> bool mask [B, H0*W0]: (method's survivors) & (conf > test_thr)."""
        B, N = next_conf_c01.shape
        h0, w0 = int(axes_lengths["h0c"]), int(axes_lengths["w0c"])
        h1, w1 = int(axes_lengths["h1c"]), int(axes_lengths["w1c"])
        idx = next_idx_c01.contiguous()
        conf = next_conf_c01.contiguous().float()
        out = ops.nms_select(conf, idx, idx, (h0, w0), (h1, w1), nms_window=self.nms_window, test_thr=float(test_thr),
                             double_check=False, extra_keep=self.extra_mask(conf, (h0, w0)))
        return out["keep_ws"][: B * N].view(B, N).bool()
