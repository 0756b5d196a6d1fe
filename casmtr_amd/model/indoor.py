"""The reference's indoor model (src/model/cascade_quadtree_stage3.py, config configs/model_configs/indoor/loftr_ds_quadtree_cas_stage3.py)
around the MI355X hot path: a frozen QuadTree matcher (ResNet-FPN backbone, 8 QuadTree layers, dual-softmax) plus the cascade
refinement at 1/4 -- a ladder side network on the raw images, POLA neighbourhood self-attention, CascadeQTAttB cross-attention
with the learned relative position bias (the `rel_pos` input of the cascade attention kernel), window matching without NMS --
and the 5x5 fine refinement.  Module / parameter / buffer names follow the reference (its checkpoints load unchanged).

    backbone            ResNetFPN_8_4_2, grey input        src/model/backbone/resnet_fpn.py:125-206
    ladder              Ladder_4_2, RGB input              resnet_fpn.py:209-276
    loftr_coarse        8 QuadtreeBlocks, top-k [32,16,16] src/model/modules/transformer.py:198-303
    loftr_coarse_4c     POLATransBlock / CascadeQuadtreeBlock + relative position tables
                                                           transformer.py:352-560, src/model/modules/POLAttention.py:70-332
    forward             cascade_quadtree_stage3.py:113-197
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..matching.cascade_matching import CascadeMatching
from ..matching.coarse_matching import CoarseMatching
from .casmtr4c import (CascadeQuadtreeBlock, CoarseTransformer, FinePreprocess, FineTransformer, SinePositionEncoding, UpBlock, _cv,
                       _fast, _grid, _lin, _ln, _swap_halves, _tokens, fine_matching, _CONV_DTYPE)


def indoor_4c_config():
    return dict(
        resnetfpn=dict(initial_dim=128, block_dims=[128, 196, 256], refine_dims=[64, 128, 256]), train_size=640, fine_window_size=5,
        coarse=dict(d_model=256, nhead=8, topks=[32, 16, 16], layer_names=["self", "cross"] * 4),
        coarse2=dict(d_model=128, nhead=4, layer_names=["self", "cross", "self", "cross"], window_size=5, attn_window_size=7, sr_ratio=2,
                     dilated=1, post_config={"method": None}),
        fine=dict(d_model=64, nhead=2, layer_names=["self", "cross"]),
        match_coarse=dict(thr=0.2, border_rm=0, train_coarse_percent=0.3, train_pad_num_gt_min=200, match_type="dual_softmax",
                          dsmax_temperature=0.1),
        match_cascade=dict(thr=0.0, test_thr=0.1, pre_thr=[0.2, 0.1], border_rm=1, double_check=True, train_pad_num_gt_min=8192,
                           match_type="softmax", dsmax_temperature=1.0))


# ------------------------------------------------------------------------------------------------------------ backbones
class _BasicBlock(nn.Module):   # resnet_fpn.py:16-43
    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn1, self.bn2 = nn.BatchNorm2d(cout), nn.BatchNorm2d(cout)
        self.downsample = None if stride == 1 else nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = self.bn2(self.conv2(F.relu(self.bn1(self.conv1(x)))))
        return F.relu((x if self.downsample is None else self.downsample(x)) + y)


def _res_layer(cin, cout, stride):
    return nn.Sequential(_BasicBlock(cin, cout, stride), _BasicBlock(cout, cout, 1))


_up2 = lambda t: F.interpolate(t, scale_factor=2.0, mode="bilinear", align_corners=True)


class ResNetFPN(nn.Module):   # ResNetFPN_8_4_2(is_rgb=False): -> [1/8 (256), 1/4 (196), 1/2 (128)]
    def __init__(self, cfg):
        super().__init__()
        d0, b = cfg["initial_dim"], cfg["block_dims"]
        self.conv1 = nn.Conv2d(1, d0, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(d0)
        self.layer1, self.layer2, self.layer3 = _res_layer(d0, b[0], 1), _res_layer(b[0], b[1], 2), _res_layer(b[1], b[2], 2)
        self.layer3_outconv = nn.Conv2d(b[2], b[2], 1, bias=False)
        self.layer2_outconv = nn.Conv2d(b[1], b[2], 1, bias=False)
        self.layer2_outconv2 = nn.Sequential(nn.Conv2d(b[2], b[2], 3, 1, 1, bias=False), nn.BatchNorm2d(b[2]), nn.LeakyReLU(),
                                             nn.Conv2d(b[2], b[1], 3, 1, 1, bias=False))
        self.layer1_outconv = nn.Conv2d(b[0], b[1], 1, bias=False)
        self.layer1_outconv2 = nn.Sequential(nn.Conv2d(b[1], b[1], 3, 1, 1, bias=False), nn.BatchNorm2d(b[1]), nn.LeakyReLU(),
                                             nn.Conv2d(b[1], b[0], 3, 1, 1, bias=False))

    def forward(self, x):
        if x.shape[1] == 3:
            x = 0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3]
        x = x.contiguous(memory_format=torch.channels_last)
        x1 = _cv(nn.Sequential(self.conv1, self.bn1, nn.ReLU(), self.layer1), x)
        x2 = _cv(self.layer2, x1)
        x3 = _cv(self.layer3, x2)
        x3o = _cv(self.layer3_outconv, x3)
        x2o = _cv(self.layer2_outconv2, _cv(self.layer2_outconv, x2) + _up2(x3o))
        x1o = _cv(self.layer1_outconv2, _cv(self.layer1_outconv, x1) + _up2(x2o))
        return x3o, x2o, x1o


class Ladder(nn.Module):   # Ladder_4_2(is_rgb=True, bn_fix=False): a light second backbone fed with the frozen one's 1/4 and 1/2 maps
    def __init__(self, cfg):
        super().__init__()
        b, r = cfg["block_dims"], cfg["refine_dims"]
        self.conv1 = nn.Conv2d(3, r[0], 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(r[0])
        self.layer1, self.layer2 = _res_layer(r[0], r[0], 1), _res_layer(r[0], r[1], 2)
        self.layer2_outconv = nn.Sequential(nn.Conv2d(r[1] + b[1], r[1], 1, bias=False), nn.BatchNorm2d(r[1]))
        self.layer1_outconv = nn.Conv2d(r[0] + b[0], r[1], 1, bias=False)
        self.layer1_outconv2 = nn.Sequential(nn.Conv2d(r[1], r[1], 3, 1, 1, bias=False), nn.BatchNorm2d(r[1]), nn.LeakyReLU(),
                                             nn.Conv2d(r[1], r[0], 3, 1, 1, bias=False), nn.BatchNorm2d(r[0]))

    def forward(self, x, f4, ff):
        x = x.contiguous(memory_format=torch.channels_last)
        x1 = _cv(nn.Sequential(self.conv1, self.bn1, nn.ReLU(), self.layer1), x)
        x2 = _cv(self.layer2, x1)
        x2o = _cv(self.layer2_outconv, torch.cat([x2, f4], 1))
        x1o = _cv(self.layer1_outconv2, _cv(self.layer1_outconv, torch.cat([x1, ff], 1)) + _up2(x2o))
        return x2o, x1o


# ------------------------------------------------------------------------------------------------------------ POLA
class _NeighbourWindowAttention(nn.Module):
    """queries of a ws x ws window attend to the (n x n windows) neighbourhood around it, with a learned relative position bias
    (NeighborWindowAttention, POLAttention.py:70-172; n = 3)"""

    def __init__(self, dim, ws, heads, n_win=3):
        super().__init__()
        self.ws, self.heads, self.n_win, self.scale = ws, heads, n_win, (dim // heads) ** -0.5
        span = (n_win + 1) * ws - 1
        self.relative_position_bias_table = nn.Parameter(torch.zeros(span * span, heads))
        q = torch.arange(ws)
        k = torch.arange(n_win * ws)
        qy, qx = torch.meshgrid(q, q, indexing="ij")
        ky, kx = torch.meshgrid(k, k, indexing="ij")
        dy = qy.reshape(-1, 1) - ky.reshape(1, -1) + n_win * ws - 1
        dx = qx.reshape(-1, 1) - kx.reshape(1, -1) + n_win * ws - 1
        self.register_buffer("relative_position_index", dy * span + dx)                 # [ws^2, (n ws)^2]
        self.Wq, self.Wk, self.Wv = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, xq, k0, v0):
        """xq [Bw, ws^2, C] window queries (un-projected); k0, v0 [Bw, (n ws)^2, C] neighbourhood keys / values already projected
        WITHOUT their biases (zero at padded positions).  The reference projects the 9x larger unfolded neighbourhoods with bias;
        the key bias adds q.b_k to every logit of a row (softmax-invariant) and the value bias adds b_v to every output (the
        weights sum to 1), so projecting once per token before unfolding gives the same result."""
        Bw, Nq, C = xq.shape
        h, d = self.heads, C // self.heads
        q = _lin(self.Wq, xq).view(Bw, Nq, h, d).transpose(1, 2)
        k = k0.view(Bw, -1, h, d).transpose(1, 2)
        v = v0.view(Bw, -1, h, d).transpose(1, 2)
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(Nq, -1, h).permute(2, 0, 1)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=bias.unsqueeze(0).to(q.dtype), scale=self.scale)
        o = o.transpose(1, 2).reshape(Bw, Nq, C) + self.Wv.bias
        return _lin(self.proj, o)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(dim, hidden), nn.Linear(hidden, dim)

    def forward(self, x):
        return _lin(self.fc2, F.gelu(_lin(self.fc1, x)))


class POLABlock(nn.Module):   # POLATransBlock, POLAttention.py:244-332
    def __init__(self, dim, heads, ws):
        super().__init__()
        self.ws = ws
        self.norm1, self.norm2 = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.attn = _NeighbourWindowAttention(dim, ws, heads)
        self.mlp = _Mlp(dim, 4 * dim)

    def forward(self, x, H, W):
        B, L, C = x.shape
        ws, n = self.ws, self.attn.n_win
        at = self.attn
        if _fast(x) and ws == 7 and C // at.heads == 32:
            # three projections of the un-padded tokens in one launch, then one kernel for the neighbourhood attention: no padding,
            # no 9x unfold, no [windows, heads, 49, 441] logits in HBM
            xn = _ln(self.norm1, x)
            dt = _CONV_DTYPE[0]
            if dt is None:
                q, k0, v0 = ops.linear_multi([xn, xn, xn], [at.Wq.weight, at.Wk.weight, at.Wv.weight], [at.Wq.bias, None, None])
            else:   # reduced glue precision: the projections only; the attention itself stays fp32
                xh = xn.to(dt)
                q = F.linear(xh, at.Wq.weight.to(dt), at.Wq.bias.to(dt)).float()
                k0, v0 = F.linear(xh, at.Wk.weight.to(dt)).float(), F.linear(xh, at.Wv.weight.to(dt)).float()
            a = ops.pola_attn(q, k0, v0, at.relative_position_bias_table.contiguous(), H, W, at.heads, ws, at.scale) + at.Wv.bias
            x = x + _lin(at.proj, a)
            return x + self.mlp(_ln(self.norm2, x))
        xn = _ln(self.norm1, x).view(B, H, W, C)
        pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws
        xn = F.pad(xn, (0, 0, 0, pr, 0, pb))                                           # zero padding takes part as keys, as in the reference
        Hp, Wp = H + pb, W + pr
        gh, gw = Hp // ws, Wp // ws
        xq = xn.view(B, gh, ws, gw, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B * gh * gw, ws * ws, C)
        m = (n // 2) * ws

        def neighbourhoods(t):   # [B, Hp, Wp, C] -> [B gh gw, (n ws)^2, C]: the n x n windows around every window, zero outside
            u = F.unfold(F.pad(t, (0, 0, m, m, m, m)).permute(0, 3, 1, 2), n * ws, stride=ws)      # [B, C (n ws)^2, gh gw]
            return u.permute(0, 2, 1).reshape(B * gh * gw, C, (n * ws) ** 2).permute(0, 2, 1)
        k0 = F.linear(xn, self.attn.Wk.weight)       # bias-free projections of the (padded) map, once per token
        v0 = F.linear(xn, self.attn.Wv.weight)
        a = self.attn(xq, neighbourhoods(k0), neighbourhoods(v0)).view(B, gh, gw, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
        x = x + a[:, :H, :W, :].reshape(B, L, C)
        return x + self.mlp(_ln(self.norm2, x))


# ------------------------------------------------------------------------------------------------------------ cascade transformer
class IndoorCascadeTransformer(nn.Module):
    """CascadeFeatureTransformer with 'POLA' self layers and relative_pe (transformer.py:352-560)"""

    def __init__(self, cfg):
        super().__init__()
        self.layer_names, self.ws, self.nhead, self.sr = cfg["layer_names"], cfg["window_size"], cfg["nhead"], cfg["sr_ratio"]
        r = self.ws // 2
        dy, dx = torch.meshgrid(torch.arange(-r, r + 1), torch.arange(-r, r + 1), indexing="ij")
        self.window = nn.Parameter(torch.stack([dy, dx], dim=-1).reshape(-1, 2), requires_grad=False)
        self.LB = self.ws * 2 if self.sr == 2 else self.ws * 6
        self.h_pos_bias = nn.Embedding(self.LB * 2 + self.sr, self.nhead)
        self.w_pos_bias = nn.Embedding(self.LB * 2 + self.sr, self.nhead)
        self.layers = nn.ModuleList(CascadeQuadtreeBlock(cfg["d_model"], cfg["nhead"], cfg.get("dilated", 1)) if n == "cross"
                                    else POLABlock(cfg["d_model"], cfg["nhead"], cfg.get("attn_window_size") or self.ws)
                                    for n in self.layer_names)

    def relative_pe(self, tgt_idx, tp, hw, hw_other, H):
        """get_relative_pe (:473-509): bias[b, head, fine token, candidate] from the offset between the token's position inside its
        coarse cell and the candidate's position relative to the coarse match.  tp [B, h w, ww, 2] (y, x) window cells."""
        (h, w), w1 = hw, hw_other[1]
        s = H // h
        W1 = w1 * s
        B = tgt_idx.shape[0]
        dev = tgt_idx.device
        iy, ix = torch.meshgrid(torch.arange(s, device=dev), torch.arange(s, device=dev), indexing="ij")
        src = torch.stack([ix, iy], -1).view(1, 1, s, 1, s, 2).expand(1, h, s, w, s, 2).reshape(1, h * s * w * s, 1, 2)   # (x, y) inside the cell
        tgt = torch.stack([tgt_idx % w1, torch.div(tgt_idx, w1, rounding_mode="trunc")], -1)                              # [B, hw, 2] (x, y)
        tgt = tgt.view(B, h, 1, w, 1, 2).expand(B, h, s, w, s, 2).reshape(B, -1, 2) * s + (s // 2 - 1)
        wi = tp * 2
        cand = torch.stack([(wi[..., 0] + a) * W1 + wi[..., 1] + b for a in (0, 1) for b in (0, 1)], dim=3).flatten(2)     # [B, hw, 4ww]
        cand = cand.view(B, h, 1, w, 1, -1).expand(B, h, 2, w, 2, cand.shape[-1]).reshape(B, h * 2 * w * 2, -1)           # every child of the cell
        cand = torch.stack([cand % W1, torch.div(cand, W1, rounding_mode="trunc")], -1)                                    # (x, y)
        rel = src - (tgt.unsqueeze(2) - cand + self.LB) + 2 * self.LB
        return (self.w_pos_bias(rel[..., 0]) + self.h_pos_bias(rel[..., 1])).permute(0, 3, 1, 2).contiguous()               # [B, nhead, HW, 4ww]

    def forward(self, f0, f1, next_idx_c01, next_idx_c10, hw8_0, hw8_1):
        (H0, W0), (H1, W1) = f0.shape[2:], f1.shape[2:]
        f0, f1 = _tokens(f0).contiguous(), _tokens(f1).contiguous()
        tp01 = ops.window_warp_idx(next_idx_c01.contiguous(), H0 // 2, W0 // 2, self.ws)
        tp10 = ops.window_warp_idx(next_idx_c10.contiguous(), H1 // 2, W1 // 2, self.ws)
        rp01 = self.relative_pe(next_idx_c01, tp01, hw8_0, hw8_1, H0)
        rp10 = self.relative_pe(next_idx_c10, tp10, hw8_1, hw8_0, H1)
        for layer, name in zip(self.layers, self.layer_names):
            if name == "self":
                f0, f1 = layer(f0, H0, W0), layer(f1, H1, W1)
            else:
                f0, f1 = layer(f0, f1, H0, W0, H1, W1, tp01, rel_pos=rp01), layer(f1, f0, H1, W1, H0, W0, tp10, rel_pos=rp10)
        return (f0.contiguous(), f1.contiguous(), ops.WindowIndex(tp01, (H0, W0), (H1, W1), 1), ops.WindowIndex(tp10, (H1, W1), (H0, W0), 1))


# ------------------------------------------------------------------------------------------------------------ the model
class CasMTRIndoor4c(nn.Module):
    def __init__(self, config=None, conv_dtype=None):
        super().__init__()
        c = self.config = config or indoor_4c_config()
        self.conv_dtype = conv_dtype
        r, ts = c["resnetfpn"]["refine_dims"], c["train_size"]
        self.backbone = ResNetFPN(c["resnetfpn"])
        self.pos_encoding = SinePositionEncoding(c["coarse"]["d_model"], (480 // 8, 640 // 8))   # hard-wired to ScanNet frames (:88)
        self.loftr_coarse = CoarseTransformer(c["coarse"])
        self.coarse_matching = CoarseMatching(c["match_coarse"], c["coarse"], materialize_conf=False,
                                              gemm=c["match_coarse"].get("gemm", "split"))   # config knob; see ops.ds_gemm_mode
        self.ladder = Ladder(c["resnetfpn"])
        self.pos_encoding_4c = SinePositionEncoding(r[1], (ts // 4, ts // 4))
        self.up_block1 = UpBlock(r[2], r[1])
        self.loftr_coarse_4c = IndoorCascadeTransformer(c["coarse2"])
        self.cascade_matching_4c = CascadeMatching(c["match_cascade"], {"propagation": "window", "dilated": 1,
                                                                       "post_config": c["coarse2"]["post_config"]}, stage="4c",
                                                   materialize_idx=False)
        self.cas_fine_preprocess = FinePreprocess(c["coarse2"]["d_model"], c["fine"]["d_model"], c["fine_window_size"], True)
        self.cas_loftr_fine = FineTransformer(c["fine"])

    def load_state_dict(self, state_dict, *args, **kwargs):
        sd = {(k[len("matcher."):] if k.startswith("matcher.") else k): v for k, v in state_dict.items()}
        return super().load_state_dict(sd, *args, **kwargs)

    def features(self, data):
        im0, im1 = data["image0"], data["image1"]
        if im0.shape != im1.shape:
            raise ValueError("the indoor model batches the two images of a pair through its ladder: equal image sizes required (:166)")
        bs = im0.shape[0]
        x = torch.cat([im0, im1], 0)
        f8, f4, ff = self.backbone(x)
        data.update(bs=bs, hw0_i=tuple(im0.shape[2:]), hw1_i=tuple(im1.shape[2:]))
        for lv, f in (("c", f8), ("8c", f8), ("4c", f4), ("2c", ff), ("f", ff)):
            data[f"hw0_{lv}"] = data[f"hw1_{lv}"] = tuple(f.shape[2:])
        return x, f8, f4, ff

    def coarse_stage(self, f8, data):
        bs = data["bs"]
        t0, t1 = self.loftr_coarse(self.pos_encoding(f8[:bs]), self.pos_encoding(f8[bs:]))
        self.coarse_matching(t0.float(), t1.float(), data, level="8c")
        return t0, t1

    def cascade_stage(self, x, f4, ff, t8_0, t8_1, data):
        bs = data["bs"]
        f4, ff = self.ladder(x, f4, ff)
        f4_0 = self.up_block1(f4[:bs], _grid(t8_0, *data["hw0_8c"]))
        f4_1 = self.up_block1(f4[bs:], _grid(t8_1, *data["hw1_8c"]))
        st8 = data["stage_8c"]
        t0, t1, idx01, idx10 = self.loftr_coarse_4c(self.pos_encoding_4c(f4_0), self.pos_encoding_4c(f4_1), st8["next_idx_c01"],
                                                    st8["next_idx_c10"], data["hw0_8c"], data["hw1_8c"])
        self.cascade_matching_4c(t0.float(), t1.float(), idx01, idx10, data, level="4c", pre_level="8c")
        return t0, t1, ff[:bs], ff[bs:]

    def fine_stage(self, ff0, ff1, t4_0, t4_1, data):
        st = data["stage_4c"]
        w0, w1 = self.cas_fine_preprocess(ff0, ff1, t4_0, t4_1, st, data["hw0_f"][0] // data["hw0_4c"][0], data["hw0_4c"][1], data["hw1_4c"][1])
        if w0.shape[0]:
            w0, w1 = self.cas_loftr_fine(w0, w1)
        scale = data["hw0_i"][0] / data["hw0_f"][0]
        if "scale0" in data:
            scale = scale * data["scale1"][st["b_ids"]]
        mk0, mk1, expec = fine_matching(w0.float(), w1.float(), st, scale)
        data.update(mkpts0_f=mk0, mkpts1_f=mk1, expec_f=expec, m_bids=st["m_bids"])
        return data

    @torch.no_grad()
    def forward(self, data):
        H, W = data["image0"].shape[2:]
        if H % 32 or W % 32:
            raise ValueError("image sides must be multiples of 32, as in the reference")
        _CONV_DTYPE[0] = self.conv_dtype
        try:
            x, f8, f4, ff = self.features(data)
            t8_0, t8_1 = self.coarse_stage(f8, data)
            t4_0, t4_1, ff0, ff1 = self.cascade_stage(x, f4, ff, t8_0, t8_1, data)
            return self.fine_stage(ff0, ff1, t4_0, t4_1, data)
        finally:
            _CONV_DTYPE[0] = None
