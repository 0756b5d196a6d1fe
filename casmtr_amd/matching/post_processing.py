"""Counterpart of src/model/functions/post_processing.py for the methods every shipped config selects:
`None` (:42-43) and 'maxpool_nms' (:111-121).  sift / softargmax_nms / d2d are kornia-based, unused by the configs
and outside the hot path: selecting them fails loudly."""
import torch

from .. import ops


class PostProcess(object):
    def __init__(self, post_config):
        self.config = post_config
        self.method = post_config["method"]
        if self.method not in (None, "maxpool_nms"):
            raise NotImplementedError(f"PostProcess method {self.method!r} is outside the MI355X hot path")
        if self.method == "maxpool_nms" and post_config.get("stride", 1) != 1:
            raise NotImplementedError("maxpool_nms is implemented for stride 1 (every shipped config)")

    @property
    def nms_window(self):
        return int(self.config["window_size"]) if self.method == "maxpool_nms" else 0

    def apply(self, data, axes_lengths, next_idx_c01, next_conf_c01, test_thr, level):
        """-> bool mask [B, H0*W0]: (NMS survivor) & (conf > test_thr)."""
        B, N = next_conf_c01.shape
        h0, w0 = int(axes_lengths["h0c"]), int(axes_lengths["w0c"])
        h1, w1 = int(axes_lengths["h1c"]), int(axes_lengths["w1c"])
        idx = next_idx_c01.contiguous()
        out = ops.nms_select(next_conf_c01.contiguous().float(), idx, idx, (h0, w0), (h1, w1), nms_window=self.nms_window,
                             test_thr=float(test_thr), double_check=False)
        return out["keep_ws"][: B * N].view(B, N).bool()
