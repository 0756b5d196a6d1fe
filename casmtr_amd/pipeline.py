"""The hot-path chain of one CasMTR forward, driven through the drop-in module surface.

This is what bench.py times and what SURVEY.md §8(d) calls a "step": for a batch of B image pairs
    2L x QTAttB.forward            (L coarse layers [self,cross]*, two directions; src/model/modules/transformer.py:294-303)
     1 x CoarseMatching.forward    (dual-softmax on the 1/8 grid; cascade_model_stage3.py:142-144)
  per cascade stage (1/4, and 1/2 for CasMTR-2c):
     2 x get_window_warp_idx       (5x5 windows around the previous stage's argmax; transformer.py:524-525)
    2c x CascadeQTAttB.forward     (c cross layers x 2 directions; transformer.py:549)
     1 x CascadeMatching.forward   (window scoring both directions, NMS + selection; cascade_model_stage3.py:167-169,
                                    cascade_model_stage4.py:194-195)
The dense q/k/v projections, MLPs, backbone and fine matching between those calls are outside the path
(SURVEY.md §2 #12-#15); their outputs are replaced by seeded synthetic tensors of the right shape.  Every attention layer
gets its OWN q/k/v tensors (as in the model, where each layer sees fresh activations), so nothing stays cache-resident
from one call to the next.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import ops
from .matching.cascade_matching import CascadeMatching
from .matching.coarse_matching import CoarseMatching
from .modules.quadtree_attention import CascadeQTAttB, QTAttB
from .modules.quadtree_block import CascadeQuadtreeAttention, QuadtreeAttention, set_caller_layout


@dataclass
class CascadeStage:
    """one CascadeFeatureTransformer + CascadeMatching pair (config['coarse2'] / ['coarse3'] + match_cascade[i])"""
    level: str = "4c"
    div: int = 4                      # grid = image / div
    dim: int = 128
    heads: int = 4
    cross_layers: int = 2             # 'cross' entries of LAYER_NAMES
    rel_pos: bool = False             # COARSE2.RELATIVE_PE (indoor): [B,nhead,H0*W0,4*ww] bias per direction
    test_thr: float = 0.2
    pre_thr: List[float] = field(default_factory=lambda: [0.2])   # one per previous stage
    border_rm: int = 2
    temperature: float = 1.0
    nms_window: int = 5               # 0 = POST_CONFIG.METHOD None


@dataclass
class HotPathConfig:
    """Default: CasMTR-4c outdoor (configs/model_configs/outdoor/loftr_ds_quadtree_cas_twins_large_stage3.py) at 832x832."""
    name: str = "CasMTR-4c outdoor 832x832"
    image_hw: Tuple[int, int] = (832, 832)
    coarse_dim: int = 256
    coarse_heads: int = 8
    coarse_topks: List[int] = field(default_factory=lambda: [32, 16, 8])
    coarse_layers: int = 6            # ['self','cross'] * 3
    window_size: int = 5
    coarse_thr: float = 0.2
    coarse_border_rm: int = 0
    coarse_temperature: float = 0.1
    stages: List[CascadeStage] = field(default_factory=lambda: [CascadeStage()])
    materialize_conf: bool = False    # data['stage_8c']['conf_matrix'] is not consumed at inference
    callers: bool = False             # enter through QuadtreeAttention / CascadeQuadtreeAttention
                                      # ([B,N,C] tokens in; q/k/v + output projections and the pyramid inside the step)
    caller_layout: str = "quads"      # callers only: the blocks' route (modules/quadtree_block.py::_quad_route); the throughput path opts into
                                      # the quad-major kernels, the nn.Module default stays token-major
    caller_gemm: str = "split"        # callers only: the blocks' projections on the f16 matrix pipe, fp32-accurate (ops.linear_gemm_mode); "exact" =
                                      # the fp32 MFMA chain
    masked: bool = False              # MegaDepth-style padding masks (BASELINE configs[2]): bottom / right up to 20 % padded
    fresh_inputs: bool = True         # every attention layer reads its own q/k/v tensors (False: one shared set, as round 1)
    paired_layers: object = "qta"     # False | True | "coarse" | "qta".  The two directions of a layer are independent in the reference
                                      # (transformer.py:295-300 / :549).  "coarse" (default): QTAttB's layout pass and coarsest level run
                                      # once for both directions on the doubled batch, the finer levels once per direction (bit-equal to
                                      # separate calls; 12.02-12.03 against 12.15-12.27 ms per step, round 4).  True: every kernel once on
                                      # the doubled batch (12.18-12.19: the gather kernels lose on 16 pairs what the coarsest level gains).
                                      # The two directions on two HIP streams instead: 565 against 588 pairs/s (round 3).  "qta" (default
                                      # since round 6): all QTAttB levels on the doubled batch, CascadeQTAttB per direction -- with dynamic item
                                      # claiming the fine levels gain on 16 pairs (2.25 -> 2.22, 1.27 -> 1.23 ms), the cascade kernel still loses
    ds_gemm: str = "split"            # CoarseMatching(gemm=...): 'split' = f16 matrix pipe + exact argmax re-decision (ops.ds_gemm_mode),
                                      # 'exact' = every logit from the fp32 chain; CASMTR_DS_GEMM overrides (bench.py's exact leg)
    implicit_windows: bool = True     # cascade window lists travel as topk_pos [B,N/4,25,2]; the int64 [B,N,100]
                                      # upsampled_idx is never written (False: the reference's data flow)

    @classmethod
    def named(cls, which: str, **kw) -> "HotPathConfig":
        """BASELINE.json configs: '4c' = configs[1]/[2], '2c' = configs[3], 'indoor' = configs[4]."""
        if which == "4c":
            return cls(**kw)
        if which == "2c":   # outdoor/loftr_ds_quadtree_cas_twins_large_stage4.py: NMS on 2c only, border_rm [1,2], pre_thr [[.2],[.2,.2]]
            return cls(name="CasMTR-2c outdoor 832x832", stages=[
                CascadeStage(level="4c", div=4, dim=128, heads=4, cross_layers=2, border_rm=1, nms_window=0),
                CascadeStage(level="2c", div=2, dim=64, heads=2, cross_layers=2, border_rm=2, pre_thr=[0.2, 0.2], nms_window=5)],
                **kw)
        if which == "indoor":   # indoor/loftr_ds_quadtree_cas_stage3.py: 640x480, topks [32,16,16], 8 coarse layers, rel_pos, no NMS
            return cls(name="CasMTR-4c indoor 640x480", image_hw=(480, 640), coarse_topks=[32, 16, 16], coarse_layers=8, stages=[
                CascadeStage(level="4c", div=4, dim=128, heads=4, cross_layers=2, rel_pos=True, test_thr=0.1, pre_thr=[0.2],
                             border_rm=1, nms_window=0)], **kw)
        raise ValueError(f"unknown config {which!r} (4c | 2c | indoor)")

    def hw(self, div):
        return self.image_hw[0] // div, self.image_hw[1] // div

    @property
    def hw8(self):
        return self.hw(8)

    @property
    def hw4(self):
        return self.hw(4)

    # first cascade stage under its round-1 names (tests, bench)
    cascade_dim = property(lambda s: s.stages[0].dim)
    cascade_heads = property(lambda s: s.stages[0].heads)
    cascade_cross_layers = property(lambda s: s.stages[0].cross_layers)
    cascade_test_thr = property(lambda s: s.stages[0].test_thr)
    cascade_pre_thr = property(lambda s: s.stages[0].pre_thr[0])
    cascade_border_rm = property(lambda s: s.stages[0].border_rm)
    cascade_temperature = property(lambda s: s.stages[0].temperature)
    nms_window = property(lambda s: s.stages[0].nms_window)


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


def _padding_masks(B, hw_i, gen, device):
    """[B,H,W] bool, True = image, False = padding: each image keeps 80-100 % of one side (multiple of 8 pixels)"""
    H, W = hw_i
    m = torch.zeros((B, H, W), dtype=torch.bool, device=device)
    frac = 0.8 + 0.2 * torch.rand(B, generator=gen, device=device).cpu()
    side = torch.rand(B, generator=gen, device=device).cpu() < 0.5
    for b in range(B):
        vh = H if side[b] else int(H * frac[b]) // 8 * 8
        vw = int(W * frac[b]) // 8 * 8 if side[b] else W
        m[b, :vh, :vw] = True
    return m


def make_synthetic_inputs(cfg: HotPathConfig, B: int, device, seed: int = 0, channels_last: bool = False) -> Dict[str, object]:
    """Keys: c{q,k,v}{0,1}[layer] (pyramids, layer index 0 when fresh_inputs is off), {level}{q,k,v}{0,1}[layer],
    {level}rel{01,10}, feat_8c{0,1}, feat_{level}{0,1}, weight, mask*."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    h8, w8 = cfg.hw8
    rn = lambda *s: torch.randn(*s, generator=g, device=device)
    cl = (lambda t: t.contiguous(memory_format=torch.channels_last)) if channels_last else (lambda t: t)
    inp = {}
    n_c = cfg.coarse_layers if cfg.fresh_inputs else 1
    for im in (0, 1):
        for n in "qkv":
            # QuadtreeAttention.forward :78-89 (per layer: the projections of that layer's input)
            inp[f"c{n}{im}"] = [[cl(t) for t in _pyramid(rn(B, cfg.coarse_dim, h8, w8))] for _ in range(n_c)]
    for st in cfg.stages:
        h, w = cfg.hw(st.div)
        n_f = st.cross_layers if cfg.fresh_inputs else 1
        for im in (0, 1):
            for n in "qkv":
                inp[f"{st.level}{n}{im}"] = [cl(rn(B, st.dim, h, w)) for _ in range(n_f)]   # CascadeQuadtreeAttention.forward
        if st.rel_pos:   # get_relative_pe output, transformer.py:499-509
            K = 4 * cfg.window_size ** 2
            inp[f"{st.level}rel01"], inp[f"{st.level}rel10"] = rn(B, st.heads, h * w, K), rn(B, st.heads, h * w, K)
    if cfg.callers:   # what LocalFeatureTransformer / CascadeFeatureTransformer hand to their attention blocks
        for im in (0, 1):
            inp[f"cx{im}"] = rn(B, h8 * w8, cfg.coarse_dim)
            for st in cfg.stages:
                h, w = cfg.hw(st.div)
                inp[f"{st.level}x{im}"] = rn(B, h * w, st.dim)
    inp["weight"] = rn(3)
    inp["feat_8c0"] = rn(B, h8 * w8, cfg.coarse_dim)
    inp["feat_8c1"] = _warp_tokens(inp["feat_8c0"], (h8, w8), (3, 5), 0.35, g)
    for st in cfg.stages:   # image 1 = image 0 moved by (24, 40) pixels at every level
        h, w = cfg.hw(st.div)
        f0 = 3.0 * rn(B, h * w, st.dim)
        inp[f"feat_{st.level}0"] = f0
        inp[f"feat_{st.level}1"] = _warp_tokens(f0, (h, w), (24 // st.div + (1 if st.div == 4 else 0), 40 // st.div), 1.0, g)
    if cfg.masked:
        inp["mask0_origin"] = _padding_masks(B, cfg.image_hw, g, device)
        inp["mask1_origin"] = _padding_masks(B, cfg.image_hw, g, device)
        for lvl, div in [("8c", 8)] + [(st.level, st.div) for st in cfg.stages]:   # set_stage_mask, cascade_model_stage3.py:60-68
            for im in (0, 1):
                inp[f"mask_{lvl}{im}"] = F.interpolate(inp[f"mask{im}_origin"].unsqueeze(1).float(), size=cfg.hw(div),
                                                        mode="nearest")[:, 0].bool().contiguous()
    return inp


class HotPath(torch.nn.Module):
    def __init__(self, cfg: HotPathConfig):
        super().__init__()
        self.cfg = cfg
        self.qta = QTAttB(cfg.coarse_heads, cfg.coarse_dim // cfg.coarse_heads, scale=3, topks=cfg.coarse_topks)
        self.cascade_qta = torch.nn.ModuleList(CascadeQTAttB(st.heads, st.dim // st.heads, dilated=1) for st in cfg.stages)
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
                torch.nn.ModuleList(unit_gain(CascadeQuadtreeAttention(st.dim, st.heads)) for _ in range(st.cross_layers))
                for st in cfg.stages)
            set_caller_layout(self, cfg.caller_layout, cfg.caller_gemm)
        self.coarse_matching = CoarseMatching(
            {"thr": cfg.coarse_thr, "border_rm": cfg.coarse_border_rm, "train_coarse_percent": 0.3,
             "train_pad_num_gt_min": 200, "match_type": "dual_softmax", "dsmax_temperature": cfg.coarse_temperature},
            materialize_conf=cfg.materialize_conf, defer_sync=True, gemm=cfg.ds_gemm)  # one host sync per step, at the end
        self.cascade_matching = torch.nn.ModuleList()
        for st in cfg.stages:
            post = {"method": "maxpool_nms", "window_size": st.nms_window} if st.nms_window else {"method": None}
            self.cascade_matching.append(CascadeMatching(
                {"thr": 0.2, "test_thr": st.test_thr, "pre_thr": list(st.pre_thr), "border_rm": st.border_rm,
                 "double_check": True, "train_pad_num_gt_min": 200, "match_type": "softmax",
                 "dsmax_temperature": st.temperature},
                {"propagation": "window", "dilated": 1, "post_config": post}, stage=st.level,
                defer_sync=True, materialize_idx=not cfg.implicit_windows))
        self.eval()

    @torch.no_grad()
    def forward(self, inp, finalize: bool = True) -> Dict[str, object]:
        """finalize=False: everything is enqueued, nothing is read back (no host sync); call finalize(out) later -- lets a driver
        keep a second batch in flight on another stream while this one's match counts travel to the host"""
        cfg = self.cfg
        h8, w8 = cfg.hw8
        data = {"hw0_i": cfg.image_hw, "hw1_i": cfg.image_hw, "hw0_8c": (h8, w8), "hw1_8c": (h8, w8)}
        for st in cfg.stages:
            data[f"hw0_{st.level}"] = data[f"hw1_{st.level}"] = cfg.hw(st.div)
        masks = {}
        if cfg.masked:
            for lvl in ["8c"] + [st.level for st in cfg.stages]:
                data[f"mask_{lvl}0"], data[f"mask_{lvl}1"] = inp[f"mask_{lvl}0"], inp[f"mask_{lvl}1"]
                masks[lvl] = (inp[f"mask_{lvl}0"].flatten(-2), inp[f"mask_{lvl}1"].flatten(-2))
        # 1. coarse transformer: the QuadTreeAttention calls of LocalFeatureTransformer.forward
        msgs = []
        for layer in range(cfg.coarse_layers):
            li = layer if cfg.fresh_inputs else 0
            pairs = ((0, 0), (1, 1)) if layer % 2 == 0 else ((0, 1), (1, 0))   # 'self' / 'cross'
            if not cfg.callers and cfg.paired_layers:
                # the two directions of a layer are independent in the reference (transformer.py:295-300): shared launches
                msgs += self.qta.forward_multi([(inp[f"cq{a}"][li], inp[f"ck{b}"][li], inp[f"cv{b}"][li]) for a, b in pairs],
                                               split_fine=cfg.paired_layers == "coarse")
                continue
            if cfg.callers and cfg.paired_layers:   # the block's own paired form: one projection launch, doubled-batch attention, one merge
                msgs += self.coarse_blocks[layer].forward_multi([(inp[f"cx{a}"], inp[f"cx{b}"]) for a, b in pairs], h8, w8)
                continue
            for a, b in pairs:
                if cfg.callers:
                    msgs.append(self.coarse_blocks[layer](inp[f"cx{a}"], inp[f"cx{b}"], h8, w8))
                else:
                    msgs.append(self.qta(inp[f"cq{a}"][li], inp[f"ck{b}"][li], inp[f"cv{b}"][li]))
        # 2. coarse matching
        m8 = masks.get("8c", (None, None))
        self.coarse_matching(inp["feat_8c0"], inp["feat_8c1"], data, mask_c0=m8[0], mask_c1=m8[1], level="8c")
        prev, pre_levels = "8c", []
        # 3. cascade stages
        for si, st in enumerate(cfg.stages):
            lvl = st.level
            h, w = cfg.hw(st.div)
            hp, wp = h // 2, w // 2
            pst = data[f"stage_{prev}"]
            tp01 = ops.window_warp_idx(pst["next_idx_c01"], hp, wp, cfg.window_size)   # transformer.py:416-440
            tp10 = ops.window_warp_idx(pst["next_idx_c10"], hp, wp, cfg.window_size)
            rel01, rel10 = inp.get(f"{lvl}rel01"), inp.get(f"{lvl}rel10")
            idx01 = idx10 = None
            for layer in range(st.cross_layers):
                li = layer if cfg.fresh_inputs else 0
                want_idx = (not cfg.implicit_windows) and layer == st.cross_layers - 1   # all layers return the same list
                if cfg.callers and cfg.paired_layers and not want_idx and rel01 is None and rel10 is None:
                    # (entering through the blocks, pairing the cascade layers too is worth 1.4 %: 604 -> 612.5 pairs/s alternating on one box --
                    #  fewer, fuller projection launches; the attention kernel alone measured the same either way, see below)
                    m0, m1 = self.cascade_blocks[si][layer].forward_multi([(inp[f"{lvl}x0"], inp[f"{lvl}x1"], tp01),
                                                                             (inp[f"{lvl}x1"], inp[f"{lvl}x0"], tp10)], h, w)
                    i01 = i10 = None
                elif cfg.callers:
                    blk = self.cascade_blocks[si][layer]
                    m0, i01 = blk(inp[f"{lvl}x0"], inp[f"{lvl}x1"], h, w, idx=tp01, rel_pos=rel01, want_idx=want_idx)
                    m1, i10 = blk(inp[f"{lvl}x1"], inp[f"{lvl}x0"], h, w, idx=tp10, rel_pos=rel10, want_idx=want_idx)
                elif cfg.paired_layers is True and not want_idx and rel01 is None and rel10 is None:   # ("coarse": a grouped layout pass
                    # with one attention launch per direction, forward_multi(split_attn=True), measured no different: 12.21 ms either way)
                    m0, m1 = self.cascade_qta[si].forward_multi([(inp[f"{lvl}q0"][li], inp[f"{lvl}k1"][li], inp[f"{lvl}v1"][li], tp01),
                                                                 (inp[f"{lvl}q1"][li], inp[f"{lvl}k0"][li], inp[f"{lvl}v0"][li], tp10)],
                                                                split_attn=False)
                    i01 = i10 = None
                else:
                    att = self.cascade_qta[si]
                    m0, i01 = att(inp[f"{lvl}q0"][li], inp[f"{lvl}k1"][li], inp[f"{lvl}v1"][li], tp01, rel01, want_idx=want_idx)
                    m1, i10 = att(inp[f"{lvl}q1"][li], inp[f"{lvl}k0"][li], inp[f"{lvl}v0"][li], tp10, rel10, want_idx=want_idx)
                idx01, idx10 = (i01, i10) if want_idx else (idx01, idx10)
                msgs += [m0, m1]
            if cfg.implicit_windows:
                idx01 = ops.WindowIndex(tp01, (h, w), (h, w), 1)
                idx10 = ops.WindowIndex(tp10, (h, w), (h, w), 1)
            pre_levels.append(prev)
            mk = masks.get(lvl, (None, None))
            self.cascade_matching[si](inp[f"feat_{lvl}0"], inp[f"feat_{lvl}1"], idx01, idx10, data, mask_c0=mk[0], mask_c1=mk[1],
                                      level=lvl, pre_level=pre_levels[0] if len(pre_levels) == 1 else list(pre_levels))
            prev = lvl
        out = {"messages": msgs, "data": data}
        return self.finalize(out) if finalize else out

    def finalize(self, out) -> Dict[str, object]:
        """The step's read-back: every stage's match count comes to the host in ONE transfer (the reference syncs per stage, at
        coarse_matching.py:126 and cascade_matching.py:254-258), then the match lists are cut to length.  The 8c stage's filter of
        padded ground-truth entries (mconf != 0, coarse_matching.py:143-151) is the one further sync."""
        data = out["data"]
        levels = [st.level for st in self.cfg.stages] + ["8c"]
        pend = [data[f"stage_{lv}"].get("_pending") for lv in levels]
        ns = [None] * len(levels)
        live = [i for i, p in enumerate(pend) if p is not None]
        if live:
            counts = torch.cat([(pend[i][0] if isinstance(pend[i], tuple) else pend[i])["n"] for i in live]).tolist()
            for i, c in zip(live, counts):
                ns[i] = c
        for lv, n in zip(levels[:-1], ns[:-1]):
            CascadeMatching.finalize(data, lv, n=n)
        CoarseMatching.finalize(data, "8c", n=ns[-1])
        last = data[f"stage_{self.cfg.stages[-1].level}"]
        out.update(m_bids=last["m_bids"], mkpts0=last["mkpts0_c"], mkpts1=last["mkpts1_c"], mconf=last["mconf"],
                   n_coarse=data["stage_8c"]["b_ids"].numel())
        return out


class RunAhead:
    """Keeps the launch stream one step ahead of the read-backs.

    `submit(inp)` enqueues step k+1 on the current stream and then finalises step k on a SIDE stream that waits only for step k's
    end-of-step event.  On one stream the read-back of step k (its .tolist() / nonzero syncs) would queue behind step k+1's kernels,
    so the host would come back only when step k+1 had finished and the GPU would idle while step k+2 was being enqueued; with the
    side stream the host is back after a few tiny kernels and step k+2 is enqueued while step k+1 still runs.  Steps themselves
    never overlap: all of a step's hot-path kernels stay on the one launch stream, in order.

    The finalised lists are allocated from the side stream's pool and consumed on the launch stream (e.g. MatchGatherer's
    collectives).  They are NOT `record_stream`-ed: the allocator would record one event on the launch stream per tensor when it is
    freed (7 per step, 5.6 us of idle stream each: the 40 us hole at every step boundary in a rocprofv3 trace, tools/trace_gaps.py).
    Instead every finalize waits for TWO launch-stream events: the end-of-step event of the step it finalises, and a "consumers"
    event recorded at the START of the submit / drain that runs it, i.e. after everything the (single) host thread had enqueued on
    the launch stream up to that call -- including every consumer of an earlier result whose blocks this finalize may be handed by
    the side pool (a caller may drop a result right after enqueueing its consumer: `consume(ra.submit(x))`, MatchGatherer.flush).
    The consumers event precedes the new step's kernels, so the wait costs no overlap.  The host waits for the side stream before
    it returns, so nothing is freed while the side stream still works on it."""

    def __init__(self, model: "HotPath"):
        self.model = model
        self.side = torch.cuda.Stream()
        self.pend = None

    def submit(self, inp):
        """-> finalised output of the PREVIOUS submit (None on the first call)"""
        consumers = torch.cuda.Event()
        consumers.record()   # launch stream, BEFORE the new step: covers every consumer of earlier results enqueued so far
        new = self.model(inp, finalize=False)
        ev = torch.cuda.Event()
        ev.record()
        old, self.pend = self.pend, (new, ev)
        return self._finish(old, consumers)

    def drain(self):
        """-> finalised output of the last submit (None if there is none)"""
        consumers = torch.cuda.Event()
        consumers.record()
        old, self.pend = self.pend, None
        return self._finish(old, consumers)

    def _finish(self, old, consumers):
        if old is None:
            return None
        out, ev = old
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            self.side.wait_event(consumers)
            res = self.model.finalize(out)
        # Everything the side stream read (step k's buffers, owned by the launch stream's pool) must be done before those buffers
        # can be dropped; the tail after finalize's last sync is a handful of gathers.
        self.side.synchronize()
        return res


# ----------------------------------------------------------------------------------------------- algorithmic work
def algorithmic_work(cfg: HotPathConfig) -> Dict[str, object]:
    """Compulsory bytes / flops per image pair, used for roofline.achieved.

    `*_bytes` follow SURVEY.md section 8(d)'s formulas literally (the reference's data flow).  With implicit windows (the default
    data flow here) two of those terms are bytes that no longer exist: the int64 `upsampled_idx` [N, 4ww] is neither written by
    CascadeQTAttB nor read by CascadeMatching (both address their candidates from topk_pos), and the 1 -> 0 direction of
    CascadeMatching writes no conf_matrix.  `*_bytes_moved` drop them, so a kernel's fraction of the HBM roof can be quoted against
    the bytes it actually has to move (`frac_moved_bytes`) next to the survey's figure (`frac_survey_formula`)."""
    h8, w8 = cfg.hw8
    N0, N1, N2 = h8 * w8, (h8 // 2) * (w8 // 2), (h8 // 4) * (w8 // 4)
    C = cfg.coarse_dim
    D = C // cfg.coarse_heads
    KW = cfg.window_size ** 2
    k1, k0 = 4 * cfg.coarse_topks[0], 4 * cfg.coarse_topks[1]
    qta_bytes = 4 * C * (3 * (N0 + N1 + N2) + N0)
    qta_flops = 2 * 2 * cfg.coarse_heads * D * (N2 * N2 + N1 * k1 + N0 * k0)
    coarse_bytes = 8 * N0 * C + 48 * N0
    coarse_flops = 2.0 * N0 * N0 * C
    calls_q = 2 * cfg.coarse_layers
    out = dict(qta_bytes=qta_bytes, qta_flops=qta_flops, coarse_bytes=coarse_bytes, coarse_flops=coarse_flops, stages={},
               fine0_bytes=4 * C * (3 * N0 + N0 + N1),    # finest level: q, k, v in, message out, parent message in
               fine1_bytes=4 * C * (3 * N1 + N1 + N2))    # middle level
    total_b, total_f = calls_q * qta_bytes + coarse_bytes, calls_q * qta_flops + coarse_flops
    total_moved = total_b
    for st in cfg.stages:
        h, w = cfg.hw(st.div)
        N, Cf, K = h * w, st.dim, 4 * KW
        rel = 4 * st.heads * N * K if st.rel_pos else 0
        pos = 8 * (N // 4) * KW * 2
        cas_bytes = 4 * Cf * 4 * N + pos + 8 * N * K + rel
        cas_moved = 4 * Cf * 4 * N + pos + rel if cfg.implicit_windows else cas_bytes
        cas_flops = 2 * 2 * N * K * Cf
        match_bytes = 2 * (8 * N * Cf + 12 * N * K + 12 * N)
        # implicit windows: per direction q + k features (8*N*Cf), topk_pos instead of the index list, next_conf + next_idx out (12*N);
        # conf_matrix (4*N*K) in the 0 -> 1 direction only
        match_moved = 2 * (8 * N * Cf + pos + 12 * N) + 4 * N * K if cfg.implicit_windows else match_bytes
        match_flops = 2 * 2 * N * K * Cf
        out["stages"][st.level] = dict(cascade_bytes=cas_bytes, cascade_bytes_moved=cas_moved, cascade_flops=cas_flops,
                                       match_bytes=match_bytes, match_bytes_moved=match_moved, match_flops=match_flops,
                                       calls=2 * st.cross_layers)
        total_b += 2 * st.cross_layers * cas_bytes + match_bytes
        total_moved += 2 * st.cross_layers * cas_moved + match_moved
        total_f += 2 * st.cross_layers * cas_flops + match_flops
    s0 = out["stages"][cfg.stages[0].level]
    out.update(cascade_bytes=s0["cascade_bytes"], cascade_flops=s0["cascade_flops"], match_bytes=s0["match_bytes"],
               match_flops=s0["match_flops"], total_bytes=total_b, total_bytes_moved=total_moved, total_flops=total_f)
    return out
