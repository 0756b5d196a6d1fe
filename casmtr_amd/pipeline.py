"""The hot-path chain of one CasMTR forward, driven through the drop-in module surface.

This is what bench.py times and what SURVEY.md §8(d) calls a "step": for a batch of B image pairs
    12 x QTAttB.forward           (6 coarse layers [self,cross]x3, two directions; src/model/modules/transformer.py:294-303)
     1 x CoarseMatching.forward   (dual-softmax on the 1/8 grid; cascade_model_stage3.py:142-144)
     2 x get_window_warp_idx      (5x5 windows around the coarse argmax; transformer.py:524-525)
     4 x CascadeQTAttB.forward    (cascade layers [cross,self,cross,self] -> 2 cross layers x 2 directions; transformer.py:549)
     1 x CascadeMatching.forward  (window scoring both directions, NMS + selection; cascade_model_stage3.py:167-169)
The dense q/k/v projections, MLPs, backbone and fine matching between those calls are outside the path
(SURVEY.md §2 #12-#15); their outputs are replaced by seeded synthetic tensors of the right shape.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

from . import ops
from .matching.cascade_matching import CascadeMatching
from .matching.coarse_matching import CoarseMatching
from .modules.quadtree_attention import CascadeQTAttB, QTAttB
from .modules.quadtree_block import CascadeQuadtreeAttention, QuadtreeAttention


@dataclass
class HotPathConfig:
    """CasMTR-4c outdoor (configs/model_configs/outdoor/loftr_ds_quadtree_cas_twins_large_stage3.py) at 832x832."""
    name: str = "CasMTR-4c outdoor 832x832"
    image_hw: Tuple[int, int] = (832, 832)
    coarse_dim: int = 256
    coarse_heads: int = 8
    coarse_topks: List[int] = field(default_factory=lambda: [32, 16, 8])
    coarse_layers: int = 6            # ['self','cross'] * 3
    cascade_dim: int = 128
    cascade_heads: int = 4
    cascade_cross_layers: int = 2     # ['cross','self','cross','self']
    window_size: int = 5
    coarse_thr: float = 0.2
    coarse_border_rm: int = 0
    coarse_temperature: float = 0.1
    cascade_test_thr: float = 0.2
    cascade_pre_thr: float = 0.2
    cascade_border_rm: int = 2
    cascade_temperature: float = 1.0
    nms_window: int = 5
    materialize_conf: bool = False    # data['stage_8c']['conf_matrix'] is not consumed at inference
    callers: bool = False             # enter through QuadtreeAttention / CascadeQuadtreeAttention
                                      # ([B,N,C] tokens in; q/k/v + output projections and the pyramid inside the step)

    @property
    def hw8(self):
        return self.image_hw[0] // 8, self.image_hw[1] // 8

    @property
    def hw4(self):
        return self.image_hw[0] // 4, self.image_hw[1] // 4


def _pyramid(x, levels=3):
    out = [x]
    for _ in range(levels - 1):
        x = F.avg_pool2d(x, kernel_size=2, stride=2)
        out.append(x)
    return out


def _warp_tokens(f0, hw, shift, noise, gen):
    """feat1[y+dy, x+dx] = feat0[y, x] + noise ; cells without a source stay random (image 1 = moved image 0)."""
    B, N, C = f0.shape
    h, w = hw
    dy, dx = shift
    f1 = torch.randn(f0.shape, generator=gen, device=f0.device)
    src = f0.view(B, h, w, C)[:, : h - dy, : w - dx]
    f1.view(B, h, w, C)[:, dy:, dx:] = src + noise * torch.randn(src.shape, generator=gen, device=f0.device)
    return f1.contiguous()


def make_synthetic_inputs(cfg: HotPathConfig, B: int, device, seed: int = 0, channels_last: bool = False) -> Dict[str, object]:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    h8, w8 = cfg.hw8
    h4, w4 = cfg.hw4
    rn = lambda *s: torch.randn(*s, generator=g, device=device)
    inp = {}
    for im in (0, 1):
        for n in "qkv":
            inp[f"c{n}{im}"] = _pyramid(rn(B, cfg.coarse_dim, h8, w8))       # QuadtreeAttention.forward :78-89
            inp[f"f{n}{im}"] = rn(B, cfg.cascade_dim, h4, w4)                 # CascadeQuadtreeAttention.forward
    if channels_last:   # same [B,C,H,W] tensors, NHWC in memory (the layout MIOpen convolutions produce when asked to)
        for key in list(inp):
            v = inp[key]
            inp[key] = [t.contiguous(memory_format=torch.channels_last) for t in v] if isinstance(v, list) else v.contiguous(memory_format=torch.channels_last)
    if cfg.callers:   # what LocalFeatureTransformer / CascadeFeatureTransformer hand to their attention blocks
        for im in (0, 1):
            inp[f"cx{im}"] = rn(B, h8 * w8, cfg.coarse_dim)
            inp[f"fx{im}"] = rn(B, h4 * w4, cfg.cascade_dim)
    inp["weight"] = rn(3)
    inp["feat_8c0"] = rn(B, h8 * w8, cfg.coarse_dim)
    inp["feat_8c1"] = _warp_tokens(inp["feat_8c0"], (h8, w8), (3, 5), 0.35, g)
    inp["feat_4c0"] = 3.0 * rn(B, h4 * w4, cfg.cascade_dim)
    inp["feat_4c1"] = _warp_tokens(inp["feat_4c0"], (h4, w4), (7, 10), 1.0, g)
    return inp


class HotPath(torch.nn.Module):
    def __init__(self, cfg: HotPathConfig):
        super().__init__()
        self.cfg = cfg
        self.qta = QTAttB(cfg.coarse_heads, cfg.coarse_dim // cfg.coarse_heads, scale=3, topks=cfg.coarse_topks)
        self.cascade_qta = CascadeQTAttB(cfg.cascade_heads, cfg.cascade_dim // cfg.cascade_heads, dilated=1)
        if cfg.callers:
            g = torch.Generator().manual_seed(1234)

            def unit_gain(m):   # random-init weights scaled so that q/k keep unit variance (trained-model-like softmaxes)
                for lin in (m.q_proj, m.k_proj, m.v_proj, m.proj):
                    with torch.no_grad():
                        lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) / lin.weight.shape[1] ** 0.5)
                return m

            self.coarse_blocks = torch.nn.ModuleList(
                unit_gain(QuadtreeAttention(cfg.coarse_dim, cfg.coarse_heads, cfg.coarse_topks, scale=3))
                for _ in range(cfg.coarse_layers))
            self.cascade_blocks = torch.nn.ModuleList(
                unit_gain(CascadeQuadtreeAttention(cfg.cascade_dim, cfg.cascade_heads))
                for _ in range(cfg.cascade_cross_layers))
        self.coarse_matching = CoarseMatching(
            {"thr": cfg.coarse_thr, "border_rm": cfg.coarse_border_rm, "train_coarse_percent": 0.3,
             "train_pad_num_gt_min": 200, "match_type": "dual_softmax", "dsmax_temperature": cfg.coarse_temperature},
            materialize_conf=cfg.materialize_conf, defer_sync=True)  # one host sync per step, at the end
        post = {"method": "maxpool_nms", "window_size": cfg.nms_window} if cfg.nms_window else {"method": None}
        self.cascade_matching = CascadeMatching(
            {"thr": 0.2, "test_thr": cfg.cascade_test_thr, "pre_thr": [cfg.cascade_pre_thr],
             "border_rm": cfg.cascade_border_rm, "double_check": True, "train_pad_num_gt_min": 200,
             "match_type": "softmax", "dsmax_temperature": cfg.cascade_temperature},
            {"propagation": "window", "dilated": 1, "post_config": post}, stage="4c")
        self.eval()

    @torch.no_grad()
    def forward(self, inp) -> Dict[str, object]:
        cfg = self.cfg
        h8, w8 = cfg.hw8
        h4, w4 = cfg.hw4
        data = {"hw0_i": cfg.image_hw, "hw1_i": cfg.image_hw, "hw0_8c": (h8, w8), "hw1_8c": (h8, w8),
                "hw0_4c": (h4, w4), "hw1_4c": (h4, w4)}
        # 1. coarse transformer: the QuadTreeAttention calls of LocalFeatureTransformer.forward
        msgs = []
        for layer in range(cfg.coarse_layers):
            if layer % 2 == 0:   # 'self'
                pairs = ((0, 0), (1, 1))
            else:                # 'cross'
                pairs = ((0, 1), (1, 0))
            for a, b in pairs:
                if cfg.callers:
                    msgs.append(self.coarse_blocks[layer](inp[f"cx{a}"], inp[f"cx{b}"], h8, w8))
                else:
                    msgs.append(self.qta(inp[f"cq{a}"], inp[f"ck{b}"], inp[f"cv{b}"]))
        # 2. coarse matching
        self.coarse_matching(inp["feat_8c0"], inp["feat_8c1"], data, level="8c")
        st8 = data["stage_8c"]
        # 3. 5x5 windows around the coarse matches
        tp01 = ops.window_warp_idx(st8["next_idx_c01"], h8, w8, cfg.window_size)
        tp10 = ops.window_warp_idx(st8["next_idx_c10"], h8, w8, cfg.window_size)
        # 4. cascade cross attention
        idx01 = idx10 = None
        for layer in range(cfg.cascade_cross_layers):
            if cfg.callers:
                m0, idx01 = self.cascade_blocks[layer](inp["fx0"], inp["fx1"], h4, w4, idx=tp01)
                m1, idx10 = self.cascade_blocks[layer](inp["fx1"], inp["fx0"], h4, w4, idx=tp10)
            else:
                m0, idx01 = self.cascade_qta(inp["fq0"], inp["fk1"], inp["fv1"], tp01, None)
                m1, idx10 = self.cascade_qta(inp["fq1"], inp["fk0"], inp["fv0"], tp10, None)
            msgs += [m0, m1]
        # 5. cascade matching (+ NMS / selection)
        self.cascade_matching(inp["feat_4c0"], inp["feat_4c1"], idx01, idx10, data, level="4c", pre_level="8c")
        st4 = data["stage_4c"]
        CoarseMatching.finalize(data, "8c")
        return {"messages": msgs, "data": data, "m_bids": st4["m_bids"], "mkpts0": st4["mkpts0_c"],
                "mkpts1": st4["mkpts1_c"], "mconf": st4["mconf"], "n_coarse": st8["b_ids"].numel()}


# ----------------------------------------------------------------------------------------------- algorithmic work
def algorithmic_work(cfg: HotPathConfig) -> Dict[str, float]:
    """Compulsory bytes / flops per image pair (SURVEY.md §8(d) formulas), used for roofline.achieved."""
    h8, w8 = cfg.hw8
    h4, w4 = cfg.hw4
    N0, N1, N2 = h8 * w8, (h8 // 2) * (w8 // 2), (h8 // 4) * (w8 // 4)
    C, Cf, N4, K = cfg.coarse_dim, cfg.cascade_dim, h4 * w4, 4 * cfg.window_size ** 2
    D = C // cfg.coarse_heads
    k1, k0 = 4 * cfg.coarse_topks[0], 4 * cfg.coarse_topks[1]
    qta_bytes = 4 * C * (3 * (N0 + N1 + N2) + N0)
    qta_flops = 2 * 2 * cfg.coarse_heads * D * (N2 * N2 + N1 * k1 + N0 * k0)
    coarse_bytes = 8 * N0 * C + 48 * N0
    coarse_flops = 2.0 * N0 * N0 * C
    cas_bytes = 4 * Cf * 4 * N4 + 8 * (N4 // 4) * cfg.window_size ** 2 * 2 + 8 * N4 * K
    cas_flops = 2 * 2 * N4 * K * Cf
    match_bytes = 2 * (8 * N4 * Cf + 12 * N4 * K + 12 * N4)
    match_flops = 2 * 2 * N4 * K * Cf
    calls_q, calls_c = 2 * cfg.coarse_layers, 2 * cfg.cascade_cross_layers
    return dict(qta_bytes=qta_bytes, qta_flops=qta_flops, coarse_bytes=coarse_bytes, coarse_flops=coarse_flops,
                cascade_bytes=cas_bytes, cascade_flops=cas_flops, match_bytes=match_bytes, match_flops=match_flops,
                total_bytes=calls_q * qta_bytes + coarse_bytes + calls_c * cas_bytes + match_bytes,
                total_flops=calls_q * qta_flops + coarse_flops + calls_c * cas_flops + match_flops)
