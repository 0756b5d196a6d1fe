"""Counterpart of src/model/functions/post_processing.py.

`None` (:42-43) and 'maxpool_nms' (:111-121) -- the methods every shipped config selects -- are folded into
casmtr_nms_select_fwd.  The others arrive at the same kernel as an extra keep mask built from torch ops on the GPU (none of them is
selected by a shipped config; they are here so that the whole PostProcess surface answers):
  'local_window_nms' (:76-93)   top-k per non-overlapping window;
  'd2d' (:122-144)              as many positions as the max-pool NMS keeps, but the top ones of the detector score S_d2d that
                                CascadeMatching.forward leaves in `data` (cascade_matching.py:90-104);
  'softargmax_nms' (:93-110)    kornia.geometry.ConvSoftArgmax2d (kornia 0.6.2, conv_soft_argmax2d in
                                kornia/geometry/subpix/spatial_soft_argmax.py), restated below from its published algorithm: kornia is
                                not in this image, so this method's parity is UNPINNED (tests compare it with an independent numpy
                                restatement of the same algorithm, not with kornia).
'maxpool_nms' / 'd2d' with stride != 1 compare a pooled [B,H0/s,W0/s] index map with a [1,H0,W0] coordinate map (:118-119) and die in
the reference with a broadcast RuntimeError; the same error is raised here.  'sift' (:44-75) is kornia's ScaleSpaceDetector pipeline
(scale pyramid, Hessian blobs, 3-D quadratic NMS, orientation) on the input IMAGE, far outside the matching hot path: it fails loudly."""
import torch
import torch.nn.functional as F

from .. import ops

_METHODS = (None, "maxpool_nms", "local_window_nms", "d2d", "softargmax_nms")


class PostProcess(object):
    def __init__(self, post_config):
        self.config = post_config
        self.method = post_config["method"]
        if self.method == "sift":
            raise NotImplementedError("PostProcess 'sift' is kornia's ScaleSpaceDetector on the input image (post_processing.py:44-75): "
                                      "outside the MI355X hot path")
        if self.method not in _METHODS:
            raise NotImplementedError(self.method)   # as the reference (:145-146), at construction instead of at the first call

    @property
    def nms_window(self):
        return int(self.config["window_size"]) if self.method == "maxpool_nms" else 0

    def _check_stride(self, hw0):
        stride = int(self.config.get("stride", 1))
        if stride != 1:
            ws = int(self.config["window_size"])
            ho, wo = ((n + 2 * (ws // 2) - ws) // stride + 1 for n in hw0)
            raise RuntimeError(f"The size of tensor a ({wo}) must match the size of tensor b ({hw0[1]}) at non-singleton dimension 2 "
                               f"(PostProcess {self.method!r} with stride {stride}: the reference compares the pooled {ho}x{wo} index "
                               f"map with the {hw0[0]}x{hw0[1]} coordinate map, post_processing.py:118-119 / :129-130)")

    def extra_mask(self, next_conf_c01, hw0, data=None, next_idx_c01=None, hw1=None):
        """-> bool [B, H0*W0] for the methods that are not folded into the selection kernel, else None."""
        if self.method == "maxpool_nms":
            self._check_stride(hw0)
            return None
        if self.method is None:
            return None
        B = next_conf_c01.shape[0]
        h, w = hw0
        if self.method == "local_window_nms":
            ws, topk = int(self.config["window_size"]), int(self.config["topk"])
            t = next_conf_c01.reshape(B, h // ws, ws, w // ws, ws).permute(0, 1, 3, 2, 4).reshape(B, -1, ws * ws)
            idx = torch.topk(t, k=topk, dim=2)[1]
            keep = torch.zeros_like(t, dtype=torch.bool).scatter_(2, idx, True)
            return keep.reshape(B, h // ws, w // ws, ws, ws).permute(0, 1, 3, 2, 4).reshape(B, h * w).contiguous()
        if self.method == "d2d":
            self._check_stride(hw0)
            # how many: the survivors of the plain max-pool NMS (no threshold), counted by the selection kernel itself
            idx = next_idx_c01 if next_idx_c01 is not None else torch.zeros_like(next_conf_c01, dtype=torch.int64)
            nms = ops.nms_select(next_conf_c01.contiguous().float(), idx.contiguous(), idx.contiguous(), hw0, hw1 or hw0,
                                 nms_window=int(self.config["window_size"]), test_thr=float("-inf"), double_check=False)
            num = nms["keep_ws"][: B * h * w].view(B, h * w).bool().sum(dim=1).tolist()   # host sync, as num[i].item() (:137)
            s = data["S_d2d"].reshape(B, -1)
            dw = int(data["d2d_w"])
            keep = torch.zeros((B, h * w), dtype=torch.bool, device=next_conf_c01.device)
            for b in range(B):
                top = torch.topk(s[b], k=min(s.shape[1], int(num[b])), largest=True, dim=0)[1]
                keep[b, (torch.div(top, dw, rounding_mode="floor") * 4) * (dw * 4) + (top % dw) * 4] = True
            return keep
        # softargmax_nms: conv_soft_argmax2d(window, stride, padding, temperature, normalized_coordinates=False)
        ws = int(self.config["window_size"])
        stride = int(self.config.get("stride", 1))
        temperature = float(self.config.get("temperature", 1.0))
        assert stride == 1 or stride == ws
        pad = ws // 2 if stride == 1 else 0
        x = next_conf_c01.reshape(B, 1, h, w).float()
        e = ((x - x.amax(dim=(2, 3), keepdim=True)) / temperature).exp()
        den = F.avg_pool2d(e, ws, stride=stride, padding=pad, divisor_override=1) + 1e-8
        # window offsets in kornia's normalised units (normalize_pixel_coordinates of the ws x ws grid: -1 .. 1), weighted by e
        off = torch.linspace(-1.0, 1.0, ws, device=x.device, dtype=x.dtype) if ws > 1 else torch.zeros(1, device=x.device, dtype=x.dtype)
        kx = off.view(1, 1, 1, ws).expand(1, 1, ws, ws)
        ky = off.view(1, 1, ws, 1).expand(1, 1, ws, ws)
        dxy = F.conv2d(e, torch.cat([kx, ky], 0).contiguous(), stride=stride, padding=pad) / den          # [B,2,ho,wo]: (x, y)
        # window centre in pixels (even windows: mean of the two central pixels), zero padding like kornia's conv2d of the global grid
        c1, c2 = ((ws // 2, ws // 2 + 1) if ws % 2 else (ws // 2 - 1, ws // 2 + 1))
        ck = torch.zeros((1, 1, ws, ws), device=x.device, dtype=x.dtype)
        ck[:, :, c1:c2, c1:c2] = 1.0 / float((c2 - c1) ** 2)
        gy, gx = torch.meshgrid(torch.arange(h, device=x.device, dtype=x.dtype), torch.arange(w, device=x.device, dtype=x.dtype), indexing="ij")
        cen = F.conv2d(torch.stack([gx, gy])[:, None], ck, stride=stride, padding=pad).squeeze(1)          # [2,ho,wo]
        coords = (dxy + cen[None]).round().long()
        flat = (coords[:, 0] * w + coords[:, 1]).reshape(B, -1)    # the reference's (:104): channel 0 (kornia's x) times w0c + channel 1
        return torch.zeros((B, h * w), dtype=torch.bool, device=x.device).scatter(1, flat, True)

    def apply(self, data, axes_lengths, next_idx_c01, next_conf_c01, test_thr, level):
        """-> bool mask [B, H0*W0]: (method's survivors) & (conf > test_thr)."""
        B, N = next_conf_c01.shape
        h0, w0 = int(axes_lengths["h0c"]), int(axes_lengths["w0c"])
        h1, w1 = int(axes_lengths["h1c"]), int(axes_lengths["w1c"])
        idx = next_idx_c01.contiguous()
        conf = next_conf_c01.contiguous().float()
        out = ops.nms_select(conf, idx, idx, (h0, w0), (h1, w1), nms_window=self.nms_window, test_thr=float(test_thr),
                             double_check=False, extra_keep=self.extra_mask(conf, (h0, w0), data, idx, (h1, w1)))
        return out["keep_ws"][: B * N].view(B, N).bool()
