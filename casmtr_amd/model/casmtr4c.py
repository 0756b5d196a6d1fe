"""CasMTR-4c (outdoor: Twins-large first two stages + FPN, QuadTree coarse transformer, window-propagated 1/4 cascade stage, 5x5 fine
refinement) as one nn.Module, written for inference around the hot-path modules of this package.

What is reproduced from the reference (for checkpoint compatibility: every submodule / parameter name and shape; for results: the
arithmetic of each block):
    backbone          TwinsFPN_8_4_2 + alt_gvt_large_first2_layers      src/model/backbone/twins_fpn.py:75-180, gvt.py:580-640
    pos_encoding_*    PositionEncodingSineNorm                           src/model/functions/position_encoding.py:55-85
    loftr_coarse_8c   LocalFeatureTransformer of QuadtreeBlocks          src/model/modules/transformer.py:141-303
    coarse_matching   CoarseMatching                                     -> casmtr_amd.matching (HIP)
    up_block1         UpBlock                                            src/model/cascade_model_stage3.py:25-47
    loftr_coarse_4c   CascadeFeatureTransformer ('window', 'local')      transformer.py:305-560, cascade_attention.py:97-300
    cascade_matching  CascadeMatching                                    -> casmtr_amd.matching (HIP)
    fine_preprocess / loftr_fine / fine_matching                         src/model/functions/fine_matching.py:14-140, transformer.py:97-139
    forward           the glue of cascade_model_stage3.py:104-178
The cascade stage hands its window lists to the matcher in their implicit form (ops.WindowIndex): the int64 upsampled_idx the
reference builds in every cross layer is never materialised (the matchers are built with materialize_idx=False;
data['stage_*']['idx_c01'].materialize() rebuilds it on request).
"""
import math
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..matching.cascade_matching import CascadeMatching
from ..matching.coarse_matching import CoarseMatching
from ..modules.quadtree_block import CascadeQuadtreeAttention, QuadtreeAttention


def outdoor_4c_config():
    """configs/model_configs/outdoor/loftr_ds_quadtree_cas_twins_large_stage3.py + configs/default.py, as a plain dict"""
    return dict(
        block_dims=[64, 128, 256], train_size=704, fine_window_size=5,
        coarse=dict(d_model=256, nhead=8, topks=[32, 16, 8], layer_names=["self", "cross"] * 3),
        coarse2=dict(d_model=128, nhead=4, layer_names=["cross", "self", "cross", "self"], window_size=5, attn_window_size=7,
                     dilated=1, post_config={"method": "maxpool_nms", "window_size": 5}),   # test_single_pair.py --NMS
        fine=dict(d_model=64, nhead=2, layer_names=["self", "cross"]),
        match_coarse=dict(thr=0.2, border_rm=0, train_coarse_percent=0.3, train_pad_num_gt_min=200, match_type="dual_softmax",
                          dsmax_temperature=0.1),
        match_cascade=dict(thr=0.0101, test_thr=0.2, pre_thr=[0.2], border_rm=2, double_check=True, train_pad_num_gt_min=4096,
                           match_type="softmax", dsmax_temperature=1.0))


# Optional reduced precision for the glue's CONVOLUTIONS (backbone, patch embeddings, up-sampling blocks) and MLP / local-attention
# GEMMs: the reference's test.py evaluates under fp16 autocast (pl.Trainer(precision=16), lightning_cascade.py:352).  QuadTree /
# cascade attention with their q/k/v projections, the matchers, LayerNorm and softmax stay fp32; outputs are cast back to fp32.
# Off by default (parity tests run fp32 throughout).
class _ThreadLocalSlot(threading.local):
    """`slot[0]` per thread: two models with different glue precisions may run from different host threads (two-stream drivers)"""
    value = None

    def __getitem__(self, i):
        return self.value

    def __setitem__(self, i, v):
        self.value = v


_CONV_DTYPE = _ThreadLocalSlot()


def _cv(module, x):
    dt = _CONV_DTYPE[0]
    if dt is None or not x.is_cuda:
        return module(x)
    with torch.autocast("cuda", dtype=dt):
        return module(x).float()


def _fast(x):
    """inference on the GPU: the token-major HIP element kernels apply (otherwise the same arithmetic on torch ops).  They have no
    backward, so they are used only with autograd off (forward() runs under no_grad; a submodule called directly with grad enabled
    takes the torch ops and its parameters receive gradients)"""
    return x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()


def _tokens(f):
    """[B,C,H,W] -> [B,H*W,C]; a view when f is channels_last"""
    return f.permute(0, 2, 3, 1).reshape(f.shape[0], -1, f.shape[1])


def _grid(t, H, W):
    """[B,H*W,C] tokens -> [B,C,H,W] as a channels_last view (no copy; MIOpen convolutions take it as is)"""
    return t.reshape(t.shape[0], H, W, t.shape[2]).permute(0, 3, 1, 2)


def _lin(layer, x):
    """nn.Linear on tokens: the package's fp32-MFMA NT GEMM (bias in the epilogue) for the big token matrices; with a reduced
    glue precision (see _cv) the MLP / local-attention GEMMs run in it, as nn.Linear does under the reference's autocast"""
    dt = _CONV_DTYPE[0]
    if dt is not None and x.is_cuda:
        return F.linear(x.to(dt), layer.weight.to(dt), None if layer.bias is None else layer.bias.to(dt)).float()
    if _fast(x) and x.shape[-1] % 32 == 0 and x.numel() // x.shape[-1] >= 4096:
        return ops.linear(x.contiguous(), layer.weight, layer.bias)
    return layer(x)


def _swap_halves(t):
    """[2B, ...] -> the two halves of the batch exchanged (image 0 <-> image 1 of every pair)"""
    B = t.shape[0] // 2
    return torch.cat([t[B:], t[:B]])


def _ln(norm, x, residual=None):
    """norm(x) (+ residual)"""
    if _fast(x) and x.shape[-1] % 4 == 0:
        return ops.layer_norm(x.contiguous(), norm.weight, norm.bias, norm.eps, None if residual is None else residual.contiguous())
    y = norm(x)
    return y if residual is None else residual + y


def outdoor_2c_config():
    """configs/model_configs/outdoor/loftr_ds_quadtree_cas_twins_large_stage4.py: a third stage at 1/2 resolution, NMS there only"""
    c = outdoor_4c_config()
    c["coarse2"]["post_config"] = {"method": None}
    c["coarse3"] = dict(d_model=64, nhead=2, layer_names=["cross", "self", "cross"], window_size=5, attn_window_size=7, dilated=1,
                        post_config={"method": "maxpool_nms", "window_size": 5})
    c["match_cascade"].update(border_rm=1, train_pad_num_gt_min=4096)
    c["match_cascade_2c"] = dict(thr=0.0101, test_thr=0.2, pre_thr=[0.2, 0.2], border_rm=2, double_check=True,
                                 train_pad_num_gt_min=8192, match_type="softmax", dsmax_temperature=1.0)
    c["fine_concat_coarse_feat"] = False   # the 1/2-level tokens are the fine features
    return c


# ------------------------------------------------------------------------------------------------------------ backbone
def _conv_bn(cin, cout, k, stride=1):
    return [nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False), nn.BatchNorm2d(cout)]


class _ResBlock(nn.Module):   # twins_fpn.py:45-72 (stride 1 only: the 1/2 encoder)
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1, bias=False)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn1, self.bn2 = nn.BatchNorm2d(cout), nn.BatchNorm2d(cout)
        self.shortcut = nn.Sequential(*_conv_bn(cin, cout, 1)) if cin != cout else None

    def forward(self, x):
        y = self.bn2(self.conv2(F.relu(self.bn1(self.conv1(x)))))
        return F.relu((x if self.shortcut is None else self.shortcut(x)) + y)


class _PatchEmbed(nn.Module):   # gvt.py:256-281
    def __init__(self, cin, cout, patch):
        super().__init__()
        self.patch = patch
        self.proj = nn.Conv2d(cin, cout, kernel_size=patch, stride=patch)
        self.norm = nn.LayerNorm(cout)

    def forward(self, x):
        H, W = x.shape[2] // self.patch, x.shape[3] // self.patch
        return _ln(self.norm, _tokens(_cv(self.proj, x))), (H, W)


class _PosCNN(nn.Module):   # gvt.py:397-411, stride 1
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Sequential(nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim))

    def forward(self, x, H, W):
        B, N, C = x.shape
        if _fast(x):
            return ops.dwconv3x3_tokens(x.contiguous(), self.proj[0].weight, self.proj[0].bias, H, W, add_input=True)
        f = _grid(x, H, W)
        return _tokens(self.proj(f) + f)


class _TokenMlp(nn.Module):   # fc1 -> GELU -> fc2 (gvt.py:47-63, cascade_attention.py:10-25)
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(dim, hidden), nn.Linear(hidden, dim)

    def forward(self, x):
        return _lin(self.fc2, F.gelu(_lin(self.fc1, x)))


class _WindowAttention(nn.Module):
    """multi-head attention inside non-overlapping ws x ws windows; the grid is zero-padded to a multiple of ws and padded keys
    are masked with -1000 (GroupAttention.forward_mask, gvt.py:102-133 / cascade_attention.py:124-157)"""

    def __init__(self, dim, heads, ws, qkv_bias):
        super().__init__()
        self.heads, self.ws, self.scale = heads, ws, (dim // heads) ** -0.5
        self.qkv = nn.Linear(dim, 3 * dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, H, W):
        B, N, C = x.shape
        ws, nh = self.ws, self.heads
        pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws
        # Reference quirk, reproduced: forward_mask builds the padding mask with `mask[:, -pad_b:, :].fill_(1)` and
        # `mask[:, :, -pad_r:].fill_(1)` (cascade_attention.py:135-137); a zero pad makes `-0:` select the whole mask, so when exactly
        # ONE grid side is a multiple of ws the mask is all ones, attn_mask is 0 everywhere and real queries also attend to the
        # zero-padded keys (k = v = the qkv bias).  Image sides that are multiples of 224 in one direction only (448x640, ...).
        unmasked_padding = (pr == 0) != (pb == 0)
        if _fast(x) and ws == 7 and C // nh == 32 and not unmasked_padding:   # one kernel on the un-padded tokens; no mask, no attention matrix in HBM
            return _lin(self.proj, ops.window_attn(_lin(self.qkv, x), H, W, nh, ws, self.scale))
        x = F.pad(x.view(B, H, W, C), (0, 0, 0, pr, 0, pb))
        Hp, Wp = H + pb, W + pr
        gh, gw = Hp // ws, Wp // ws
        pad = torch.zeros((1, Hp, Wp), device=x.device)
        if pb and not unmasked_padding:
            pad[:, -pb:, :] = 1
        if pr and not unmasked_padding:
            pad[:, :, -pr:] = 1
        pad = pad.reshape(1, gh, ws, gw, ws).transpose(2, 3).reshape(1, gh * gw, ws * ws)
        bias = pad.unsqueeze(2) - pad.unsqueeze(3)                        # != 0 where exactly one of (query, key) is padding
        bias = torch.where(bias != 0, torch.full_like(bias, -1000.0), torch.zeros_like(bias))
        xw = x.reshape(B, gh, ws, gw, ws, C).transpose(2, 3)             # [B, gh, gw, ws, ws, C]
        qkv = self.qkv(xw).reshape(B, gh * gw, ws * ws, 3, nh, C // nh).permute(3, 0, 1, 4, 2, 5)
        q, k, v = qkv[0], qkv[1], qkv[2]                                  # [B, windows, heads, ws*ws, d]
        att = ((q @ k.transpose(-2, -1)) * self.scale + bias.unsqueeze(2)).softmax(dim=-1)
        out = (att @ v).transpose(2, 3).reshape(B, gh, gw, ws, ws, C).transpose(2, 3).reshape(B, Hp, Wp, C)
        return self.proj(out[:, :H, :W, :].reshape(B, N, C))


class _ReducedAttention(nn.Module):
    """global attention against keys / values from a stride-sr convolution of the token grid (Attention, gvt.py:161-203)"""

    def __init__(self, dim, heads, sr, qkv_bias):
        super().__init__()
        self.heads, self.scale = heads, (dim // heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, 2 * dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.sr = nn.Conv2d(dim, dim, kernel_size=sr, stride=sr)
        self.norm = nn.LayerNorm(dim)

    def forward(self, x, H, W):
        B, N, C = x.shape
        nh = self.heads
        q = _lin(self.q, x).reshape(B, N, nh, C // nh).permute(0, 2, 1, 3)
        r = _ln(self.norm, _tokens(_cv(self.sr, _grid(x, H, W))))
        kv = self.kv(r).reshape(B, -1, 2, nh, C // nh).permute(2, 0, 3, 1, 4)
        # softmax(q k^T * scale) v without the [B, heads, N, N/sr^2] matrix in HBM (7.5 GB at 832x832, batch 8)
        o = F.scaled_dot_product_attention(q, kv[0], kv[1], scale=self.scale)
        return _lin(self.proj, o.transpose(1, 2).reshape(B, N, C))


class _TokenBlock(nn.Module):   # pre-norm transformer block on a token grid (GroupBlock, gvt.py:239-253 / cascade_attention.py:214-228)
    def __init__(self, dim, heads, attn, eps):
        super().__init__()
        self.norm1, self.norm2 = nn.LayerNorm(dim, eps=eps), nn.LayerNorm(dim, eps=eps)
        self.attn = attn
        self.mlp = _TokenMlp(dim, 4 * dim)

    def forward(self, x, H, W):
        x = x + self.attn(_ln(self.norm1, x), H, W)
        return x + self.mlp(_ln(self.norm2, x))


class _TwinsStages(nn.Module):
    """the first two stages of Twins-SVT-large (alt_gvt_large_first2_layers, gvt.py:580-640, 822-827): per stage a patch
    embedding, a window-attention block, the conditional position encoding, a reduced global-attention block, a LayerNorm.
    patch_embeds[2:] / pos_block[2:] are never run: they exist because the reference's state dict carries them."""
    DIMS, HEADS, SR = [128, 256, 512, 1024], [4, 8, 16, 32], [8, 4, 2, 1]

    def __init__(self):
        super().__init__()
        d = self.DIMS
        self.embed_dims = d[:2]
        self.patch_embeds = nn.ModuleList([_PatchEmbed(3, d[0], 4)] + [_PatchEmbed(d[i - 1], d[i], 2) for i in (1, 2, 3)])
        self.pos_block = nn.ModuleList(_PosCNN(c) for c in d)
        self.norm_list = nn.ModuleList(nn.LayerNorm(c, eps=1e-6) for c in d[:2])
        self.blocks = nn.ModuleList(
            nn.ModuleList([_TokenBlock(d[k], self.HEADS[k], _WindowAttention(d[k], self.HEADS[k], 7, True), 1e-6),
                           _TokenBlock(d[k], self.HEADS[k], _ReducedAttention(d[k], self.HEADS[k], self.SR[k], True), 1e-6)])
            for k in range(2))

    def forward_features(self, x):
        outs = []
        B = x.shape[0]
        for i in range(2):
            x, (H, W) = self.patch_embeds[i](x)
            x = self.blocks[i][0](x, H, W)
            x = self.pos_block[i](x, H, W)
            x = self.blocks[i][1](x, H, W)
            x = _grid(_ln(self.norm_list[i], x), H, W)
            outs.append(x)
        return outs


class TwinsFPN(nn.Module):   # twins_fpn.py:75-180 -> [1/8 (C=256), 1/4 (C=128), 1/2 (C=64)]
    def __init__(self, block_dims):
        super().__init__()
        b = block_dims
        self.vit = _TwinsStages()
        e = self.vit.embed_dims
        self.conv1 = nn.Sequential(nn.Conv2d(3, b[0] // 2, 7, 2, 3, bias=False), nn.BatchNorm2d(b[0] // 2), nn.ReLU(inplace=True))
        self.layer1 = nn.Sequential(_ResBlock(b[0] // 2, b[0]), _ResBlock(b[0], b[0]))
        self.layer3_outconv = nn.Sequential(*_conv_bn(e[1], b[2], 1))
        self.layer2_outconv = nn.Sequential(*_conv_bn(e[0], b[2], 1))
        self.layer2_outconv2 = nn.Sequential(*_conv_bn(b[2], b[2], 3), nn.LeakyReLU(), *_conv_bn(b[2], b[1], 3))
        self.layer1_outconv = nn.Sequential(*_conv_bn(b[0], b[1], 1))
        self.layer1_outconv2 = nn.Sequential(*_conv_bn(b[1], b[1], 3), nn.LeakyReLU(), *_conv_bn(b[1], b[0], 3))

    def forward(self, x):
        mean = x.new_tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
        std = x.new_tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        x = ((x - mean) / std).contiguous(memory_format=torch.channels_last)
        x1 = _cv(nn.Sequential(self.conv1, self.layer1), x)
        x2, x3 = self.vit.forward_features(x)
        up = lambda t: F.interpolate(t, scale_factor=2.0, mode="bilinear", align_corners=True)
        x3o = _cv(self.layer3_outconv, x3)
        x2o = _cv(self.layer2_outconv2, _cv(self.layer2_outconv, x2) + up(x3o))
        x1o = _cv(self.layer1_outconv2, _cv(self.layer1_outconv, x1) + up(x2o))
        return x3o, x2o, x1o


# ------------------------------------------------------------------------------------------------------------ transformers
class SinePositionEncoding(nn.Module):   # position_encoding.py:55-85 (positions normalised to the training grid)
    def __init__(self, d_model, max_shape):
        super().__init__()
        self.d_model, self.max_shape = d_model, max_shape
        self._pe = None

    def forward(self, x):
        H, W = x.shape[2:]
        if self._pe is None or self._pe.shape[2:] != (H, W) or self._pe.device != x.device:
            ypos = torch.ones((H, W)).cumsum(0).float().unsqueeze(0) * self.max_shape[0] / H
            xpos = torch.ones((H, W)).cumsum(1).float().unsqueeze(0) * self.max_shape[1] / W
            div = torch.exp(torch.arange(0, self.d_model // 2, 2).float() * (-math.log(10000.0) / (self.d_model // 2)))[:, None, None]
            pe = torch.zeros((self.d_model, H, W))
            pe[0::4], pe[1::4] = torch.sin(xpos * div), torch.cos(xpos * div)
            pe[2::4], pe[3::4] = torch.sin(ypos * div), torch.cos(ypos * div)
            self._pe = pe.unsqueeze(0).to(x.device).contiguous(memory_format=torch.channels_last)
        return x + self._pe


class _DWConv(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def forward(self, x, H, W):
        B, N, C = x.shape
        return _tokens(self.dwconv(_grid(x, H, W)))


class _ConvMlp(nn.Module):   # transformer.py:52-94: fc1 -> ReLU -> depth-wise 3x3 -> GELU -> fc2
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.dwconv, self.fc2 = nn.Linear(dim, hidden), _DWConv(hidden), nn.Linear(hidden, dim)

    def forward(self, x, H, W):
        h = _lin(self.fc1, x)
        if _fast(h):   # ReLU, the depth-wise convolution and GELU in one token-major pass
            return _lin(self.fc2, ops.dwconv3x3_tokens(h.contiguous(), self.dwconv.dwconv.weight, self.dwconv.dwconv.bias, H, W,
                                                       pre_relu=True, post_gelu=True))
        return self.fc2(F.gelu(self.dwconv(F.relu(h), H, W)))


class QuadtreeBlock(nn.Module):   # transformer.py:141-196
    def __init__(self, dim, heads, topks):
        super().__init__()
        self.norm1, self.norm2 = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.attn = QuadtreeAttention(dim, num_heads=heads, topks=topks, scale=3, attn_type="B")
        self.mlp = _ConvMlp(dim, 4 * dim)

    def forward(self, x, target, H, W, H1, W1, swap=False):
        """swap: x holds both images of every pair ([2B,N,C], image 0 first) and the target is the other half -- one launch per
        layer for both directions; norm1(target) is then the swapped norm1(x)"""
        xn = _ln(self.norm1, x)
        tn = _swap_halves(xn) if swap else (xn if target is x else _ln(self.norm1, target))
        x = x + self.attn(xn, tn, H, W, H1, W1)
        return x + self.mlp(_ln(self.norm2, x), H, W)


class CascadeQuadtreeBlock(nn.Module):   # transformer.py:305-345
    def __init__(self, dim, heads, dilated):
        super().__init__()
        self.norm1, self.norm2 = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.attn = CascadeQuadtreeAttention(dim, num_heads=heads, scale=2, dilated=dilated)
        self.mlp = _ConvMlp(dim, 4 * dim)

    def forward(self, x, target, H, W, H1, W1, idx, swap=False, rel_pos=None):
        xn = _ln(self.norm1, x)
        y, _ = self.attn(xn, _swap_halves(xn) if swap else _ln(self.norm1, target), H, W, H1, W1, idx, rel_pos, want_idx=False)
        x = x + y
        return x + self.mlp(_ln(self.norm2, x), H, W)


class LocalBlock(nn.Module):   # cascade_attention.py:240-248: window self-attention, ws = attn_window_size
    def __init__(self, dim, heads, ws):
        super().__init__()
        self.block_local = _TokenBlock(dim, heads, _WindowAttention(dim, heads, ws, False), 1e-5)

    def forward(self, x, H, W):
        return self.block_local(x, H, W)


class CoarseTransformer(nn.Module):   # LocalFeatureTransformer, block_type 'quadtree' (transformer.py:198-303)
    def __init__(self, cfg):
        super().__init__()
        self.layer_names = cfg["layer_names"]
        self.layers = nn.ModuleList(QuadtreeBlock(cfg["d_model"], cfg["nhead"], cfg["topks"]) for _ in self.layer_names)

    def forward(self, f0, f1):
        (H0, W0), (H1, W1) = f0.shape[2:], f1.shape[2:]
        if f0.shape == f1.shape:   # both images of the pairs in one batch: half the launches, same arithmetic per image
            B = f0.shape[0]
            x = torch.cat([_tokens(f0), _tokens(f1)])
            for layer, name in zip(self.layers, self.layer_names):
                x = layer(x, x, H0, W0, H0, W0, swap=(name != "self"))
            return x[:B], x[B:]
        f0, f1 = _tokens(f0).contiguous(), _tokens(f1).contiguous()
        for layer, name in zip(self.layers, self.layer_names):
            if name == "self":
                f0, f1 = layer(f0, f0, H0, W0, H0, W0), layer(f1, f1, H1, W1, H1, W1)
            else:
                f0, f1 = layer(f0, f1, H0, W0, H1, W1), layer(f1, f0, H1, W1, H0, W0)
        return f0, f1


class CascadeTransformer(nn.Module):   # CascadeFeatureTransformer, 'window' propagation, 'local' self-attention (transformer.py:347-560)
    def __init__(self, cfg):
        super().__init__()
        self.layer_names, self.ws, self.dilated = cfg["layer_names"], cfg["window_size"], cfg.get("dilated", 1)
        r = self.ws // 2
        dy, dx = torch.meshgrid(torch.arange(-r, r + 1), torch.arange(-r, r + 1), indexing="ij")
        self.window = nn.Parameter(torch.stack([dy, dx], dim=-1).reshape(-1, 2), requires_grad=False)   # propagations.py: (dy, dx) row-major
        self.layers = nn.ModuleList(
            CascadeQuadtreeBlock(cfg["d_model"], cfg["nhead"], self.dilated) if n == "cross"
            else LocalBlock(cfg["d_model"], cfg["nhead"], cfg.get("attn_window_size") or self.ws) for n in self.layer_names)

    def forward(self, f0, f1, next_idx_c01, next_idx_c10):
        (H0, W0), (H1, W1) = f0.shape[2:], f1.shape[2:]
        f0, f1 = _tokens(f0).contiguous(), _tokens(f1).contiguous()
        tp01 = ops.window_warp_idx(next_idx_c01.contiguous(), H0 // 2, W0 // 2, self.ws)   # get_window_warp_idx, :416-440
        tp10 = ops.window_warp_idx(next_idx_c10.contiguous(), H1 // 2, W1 // 2, self.ws)
        if f0.shape == f1.shape:   # both directions per launch
            B = f0.shape[0]
            x, tp = torch.cat([f0, f1]), torch.cat([tp01, tp10])
            for layer, name in zip(self.layers, self.layer_names):
                x = layer(x, H0, W0) if name == "self" else layer(x, x, H0, W0, H0, W0, tp, swap=True)
            return (x[:B], x[B:], ops.WindowIndex(tp01, (H0, W0), (H1, W1), self.dilated),
                    ops.WindowIndex(tp10, (H1, W1), (H0, W0), self.dilated))
        for layer, name in zip(self.layers, self.layer_names):
            if name == "self":
                f0, f1 = layer(f0, H0, W0), layer(f1, H1, W1)
            else:
                f0, f1 = layer(f0, f1, H0, W0, H1, W1, tp01), layer(f1, f0, H1, W1, H0, W0, tp10)
        return (f0.contiguous(), f1.contiguous(), ops.WindowIndex(tp01, (H0, W0), (H1, W1), self.dilated),
                ops.WindowIndex(tp10, (H1, W1), (H0, W0), self.dilated))


class UpBlock(nn.Module):   # cascade_model_stage3.py:25-47
    def __init__(self, dim1, dim2):
        super().__init__()
        self.inner = nn.Sequential(nn.Conv2d(dim1, dim2, 1, bias=False), nn.BatchNorm2d(dim2))
        self.up = nn.Sequential(nn.Conv2d(dim2, dim2, 3, padding=1, bias=False), nn.BatchNorm2d(dim2), nn.LeakyReLU())

    def forward(self, fine, coarse):
        return _cv(self.up, fine + _cv(self.inner, F.interpolate(coarse, scale_factor=2, mode="bilinear", align_corners=True)))


# ------------------------------------------------------------------------------------------------------------ fine level
class _EncoderLayer(nn.Module):   # LoFTREncoderLayer with full attention (transformer.py:97-139, linear_attention.py:52-81)
    def __init__(self, d, heads):
        super().__init__()
        self.heads = heads
        self.q_proj, self.k_proj, self.v_proj, self.merge = (nn.Linear(d, d, bias=False) for _ in range(4))
        self.mlp = nn.Sequential(nn.Linear(2 * d, 2 * d, bias=False), nn.ReLU(True), nn.Linear(2 * d, d, bias=False))
        self.norm1, self.norm2 = nn.LayerNorm(d), nn.LayerNorm(d)

    def forward(self, x, src):
        B, L, C = x.shape
        h, d = self.heads, C // self.heads
        q, k, v = self.q_proj(x).view(B, L, h, d), self.k_proj(src).view(B, -1, h, d), self.v_proj(src).view(B, -1, h, d)
        att = torch.softmax(torch.einsum("nlhd,nshd->nlsh", q, k) / d ** 0.5, dim=2)
        msg = _ln(self.norm1, self.merge(torch.einsum("nlsh,nshd->nlhd", att, v).reshape(B, L, C)))
        return _ln(self.norm2, self.mlp(torch.cat([x, msg], dim=2)), residual=x)


class FineTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer_names = cfg["layer_names"]
        self.layers = nn.ModuleList(_EncoderLayer(cfg["d_model"], cfg["nhead"]) for _ in self.layer_names)

    def forward(self, f0, f1):
        for layer, name in zip(self.layers, self.layer_names):
            if name == "self":
                f0, f1 = layer(f0, f0), layer(f1, f1)
            else:
                f0 = layer(f0, f1)
                f1 = layer(f1, f0)   # the second direction sees the updated f0 (block_type 'loftr', transformer.py:282-284)
        return f0, f1


class FinePreprocess(nn.Module):   # CascadeFinePreprocess, fine_matching.py:14-67
    def __init__(self, d_coarse, d_fine, W, cat_coarse=True):
        super().__init__()
        self.W, self.d_fine, self.cat_coarse = W, d_fine, cat_coarse
        if cat_coarse:
            self.down_proj = nn.Linear(d_coarse, d_fine)
            self.merge_feat = nn.Linear(2 * d_fine, d_fine)

    def forward(self, ff0, ff1, fc0, fc1, st, stride, wc0, wc1):
        """ff*: [B,C,H,W] fine maps; fc*: [B,hw,Cc] tokens of the level the matches live on (None without the concat); wc*: that
        level's grid width.  The reference unfolds the whole map ([B, C*W*W, L]: 2 x 8.9 GB at 832x832, batch 8, 1/2 level) and
        then indexes it; here only the matched windows are gathered -- same values, zero padding included."""
        W, r = self.W, self.W // 2
        b, i, j = st["b_ids"], st["i_ids"], st["j_ids"]
        if b.numel() == 0:
            e = torch.empty(0, W * W, self.d_fine, device=ff0.device)
            return e, e
        d = torch.arange(-r, r + 1, device=ff0.device)
        dy, dx = torch.meshgrid(d, d, indexing="ij")
        dy, dx = dy.reshape(1, -1), dx.reshape(1, -1)

        def windows(f, ids, wc):   # W x W patch of the fine map around every selected token (centre = token * stride)
            fp = F.pad(f, (r, r, r, r)).permute(0, 2, 3, 1)                                   # [B, H+2r, W+2r, C]
            ys = (torch.div(ids, wc, rounding_mode="trunc") * stride + r)[:, None] + dy
            xs = ((ids % wc) * stride + r)[:, None] + dx
            return fp[b[:, None], ys, xs]                                                     # [n, WW, C]
        w0, w1 = windows(ff0, i, wc0), windows(ff1, j, wc1)
        if not self.cat_coarse:
            return w0, w1
        c = self.down_proj(torch.cat([fc0[b, i], fc1[b, j]], 0))                              # [2n, d_fine]
        m = self.merge_feat(torch.cat([torch.cat([w0, w1], 0), c[:, None, :].expand(-1, W * W, -1)], -1))
        return torch.chunk(m, 2, dim=0)


def fine_matching(f0, f1, st, scale):
    """CascadeFineMatching (fine_matching.py:70-140): expectation of the centre feature's correlation heat map -> sub-pixel
    offset inside the W x W window.  -> (mkpts0_f, mkpts1_f, expec_f)"""
    M, WW, C = f0.shape
    if M == 0:
        return st["mkpts0_c"], st["mkpts1_c"], torch.empty(0, 3, device=f0.device)
    W = int(math.sqrt(WW))
    heat = torch.softmax(torch.einsum("mc,mrc->mr", f0[:, WW // 2, :], f1) / C ** 0.5, dim=1)
    lin = torch.linspace(-1.0, 1.0, W, device=f0.device)
    gy, gx = torch.meshgrid(lin, lin, indexing="ij")
    grid = torch.stack([gx, gy], dim=-1).reshape(1, WW, 2)               # (x, y), normalised
    coords = (heat.unsqueeze(-1) * grid).sum(dim=1)                       # spatial expectation
    var = (grid ** 2 * heat.unsqueeze(-1)).sum(dim=1) - coords ** 2
    std = torch.sqrt(torch.clamp(var, min=1e-10)).sum(-1)
    mk1 = st["mkpts1_c"] + (coords * (W // 2) * scale)[: len(st["mconf"])]
    return st["mkpts0_c"], mk1, torch.cat([coords, std[:, None]], -1)


# ------------------------------------------------------------------------------------------------------------ the model
class CasMTR4c(nn.Module):
    """CasMTR-4c (cascade_model_stage3.py); with config['coarse3'] the 1/2-resolution third stage of CasMTR-2c
    (cascade_model_stage4.py) is added: up_block2, loftr_coarse_2c, cascade_matching_2c, fine refinement on the 1/2-level tokens."""

    def __init__(self, config=None, conv_dtype=None):
        """conv_dtype: None (fp32 everywhere) | torch.float16 | torch.bfloat16 for the glue's convolutions (see _cv)"""
        super().__init__()
        c = self.config = config or outdoor_4c_config()
        self.conv_dtype = conv_dtype
        b, ts = c["block_dims"], c["train_size"]
        self.has_2c = c.get("coarse3") is not None
        self.fine_level = "2c" if self.has_2c else "4c"
        self.backbone = TwinsFPN(b)
        self.pos_encoding_8c = SinePositionEncoding(b[2], (ts // 8, ts // 8))
        self.loftr_coarse_8c = CoarseTransformer(c["coarse"])
        self.coarse_matching_8c = CoarseMatching(c["match_coarse"], c["coarse"], materialize_conf=False,
                                                 gemm=c["match_coarse"].get("gemm", "split"))   # config knob; see ops.ds_gemm_mode
        self.pos_encoding_4c = SinePositionEncoding(b[1], (ts // 4, ts // 4))
        self.up_block1 = UpBlock(b[2], b[1])
        self.loftr_coarse_4c = CascadeTransformer(c["coarse2"])
        cas = lambda cc: {"propagation": "window", "dilated": cc.get("dilated", 1), "post_config": cc["post_config"]}
        self.cascade_matching_4c = CascadeMatching(c["match_cascade"], cas(c["coarse2"]), stage="4c", materialize_idx=False)
        if self.has_2c:
            self.pos_encoding_2c = SinePositionEncoding(b[0], (ts // 2, ts // 2))
            self.up_block2 = UpBlock(b[1], b[0])
            self.loftr_coarse_2c = CascadeTransformer(c["coarse3"])
            self.cascade_matching_2c = CascadeMatching(c["match_cascade_2c"], cas(c["coarse3"]), stage="2c", materialize_idx=False)
        self.fine_preprocess = FinePreprocess(c["coarse2"]["d_model"], c["fine"]["d_model"], c["fine_window_size"],
                                              c.get("fine_concat_coarse_feat", True))
        self.loftr_fine = FineTransformer(c["fine"])

    def load_state_dict(self, state_dict, *args, **kwargs):   # cascade_model_stage3.py:180-184
        sd = {(k[len("matcher."):] if k.startswith("matcher.") else k): v for k, v in state_dict.items()}
        return super().load_state_dict(sd, *args, **kwargs)

    # the forward pass in pieces (tests drive them one at a time on the reference's stage inputs)
    def features(self, data):
        """backbone -> (f8_0, f8_1), (f4_0, f4_1), (ff0, ff1); records the grid sizes in data"""
        im0, im1 = data["image0"], data["image1"]
        bs = im0.shape[0]
        data.update(bs=bs, hw0_i=tuple(im0.shape[2:]), hw1_i=tuple(im1.shape[2:]))
        if im0.shape == im1.shape:
            f8, f4, ff = self.backbone(torch.cat([im0, im1], 0))
            out = f8.split(bs), f4.split(bs), ff.split(bs)
        else:
            a, b = self.backbone(im0), self.backbone(im1)
            out = tuple(zip(a, b))
        (f8_0, f8_1), (f4_0, f4_1), (ff0, ff1) = out
        data.update(hw0_8c=tuple(f8_0.shape[2:]), hw1_8c=tuple(f8_1.shape[2:]), hw0_4c=tuple(f4_0.shape[2:]),
                    hw1_4c=tuple(f4_1.shape[2:]), hw0_2c=tuple(ff0.shape[2:]), hw1_2c=tuple(ff1.shape[2:]),
                    hw0_f=tuple(ff0.shape[2:]), hw1_f=tuple(ff1.shape[2:]))
        return out

    def _masks(self, data, level):   # set_stage_mask, cascade_model_stage3.py:60-68
        if "mask0_origin" not in data:
            return None, None
        m = [F.interpolate(data[f"mask{i}_origin"].unsqueeze(1).float(), size=data[f"hw{i}_{level}"], mode="nearest")[:, 0].bool()
             for i in (0, 1)]
        data[f"mask_{level}0"], data[f"mask_{level}1"] = m
        return m[0].flatten(-2), m[1].flatten(-2)

    def coarse_stage(self, f8_0, f8_1, data):
        """1/8: position encoding, QuadTree transformer, dual-softmax matching -> tokens [B, hw, C] x 2; data['stage_8c']"""
        t0, t1 = self.loftr_coarse_8c(self.pos_encoding_8c(f8_0), self.pos_encoding_8c(f8_1))
        m0, m1 = self._masks(data, "8c")
        self.coarse_matching_8c(t0.float(), t1.float(), data, mask_c0=m0, mask_c1=m1, level="8c")
        return t0, t1

    def cascade_stage(self, f_0, f_1, tp_0, tp_1, data, level="4c"):
        """1/4 (or 1/2): up-sample the previous level's tokens into this level's backbone features, cascade transformer around the
        previous level's argmax, window matching (+ NMS) -> tokens [B, HW, C] x 2; data['stage_<level>']"""
        prev, pre_levels = ("8c", "8c") if level == "4c" else ("4c", ["8c", "4c"])
        up, pe, tr, mt = ((self.up_block1, self.pos_encoding_4c, self.loftr_coarse_4c, self.cascade_matching_4c) if level == "4c" else
                          (self.up_block2, self.pos_encoding_2c, self.loftr_coarse_2c, self.cascade_matching_2c))
        f_0 = up(f_0, _grid(tp_0, *data[f"hw0_{prev}"]))
        f_1 = up(f_1, _grid(tp_1, *data[f"hw1_{prev}"]))
        stp = data[f"stage_{prev}"]
        t0, t1, idx01, idx10 = tr(pe(f_0), pe(f_1), stp["next_idx_c01"], stp["next_idx_c10"])
        m0, m1 = self._masks(data, level)
        mt(t0.float(), t1.float(), idx01, idx10, data, mask_c0=m0, mask_c1=m1, level=level, pre_level=pre_levels)
        return t0, t1

    def fine_stage(self, ff0, ff1, tc_0, tc_1, data):
        """W x W refinement around every match of the last cascade level -> data['mkpts0_f' | 'mkpts1_f' | 'expec_f' | 'm_bids'].
        4c model: windows of the 1/2 backbone map + the projected 1/4 tokens; 2c model: windows of the 1/2-level tokens themselves."""
        lv = self.fine_level
        st = data[f"stage_{lv}"]
        if self.has_2c:
            ff0, ff1 = _grid(tc_0, *data["hw0_2c"]), _grid(tc_1, *data["hw1_2c"])
        w0, w1 = self.fine_preprocess(ff0, ff1, tc_0, tc_1, st, data["hw0_f"][0] // data[f"hw0_{lv}"][0], data[f"hw0_{lv}"][1],
                                      data[f"hw1_{lv}"][1])
        if w0.shape[0]:
            w0, w1 = self.loftr_fine(w0, w1)
        scale = data["hw0_i"][0] / data["hw0_f"][0]
        if "scale0" in data:   # per-pair resize factors of the dataset loaders (fine_matching.py:131)
            scale = scale * data["scale1"][st["b_ids"]]
        mk0, mk1, expec = fine_matching(w0.float(), w1.float(), st, scale)
        data.update(mkpts0_f=mk0, mkpts1_f=mk1, expec_f=expec, m_bids=st["m_bids"])
        return data

    @torch.no_grad()
    def forward(self, data):
        """data: {'image0','image1': [N,3,H,W] in [0,1]; optional 'mask0_origin','mask1_origin' [N,H,W] bool, 'scale0','scale1'}.
        Updated in place with the reference's keys: hw*_i / hw*_8c / hw*_4c / hw*_2c / hw*_f, stage_8c, stage_4c (stage_2c), m_bids,
        mkpts0_f, mkpts1_f, expec_f."""
        H, W = data["image0"].shape[2:]
        if H % 32 or W % 32 or data["image1"].shape[2] % 32 or data["image1"].shape[3] % 32:
            raise ValueError("image sides must be multiples of 32 (three 2x2 quadtree levels below the 1/8 grid), as in the reference")
        _CONV_DTYPE[0] = self.conv_dtype
        try:
            (f8_0, f8_1), (f4_0, f4_1), (ff0, ff1) = self.features(data)
            t8_0, t8_1 = self.coarse_stage(f8_0, f8_1, data)
            t_0, t_1 = self.cascade_stage(f4_0, f4_1, t8_0, t8_1, data, "4c")
            if self.has_2c:
                t_0, t_1 = self.cascade_stage(ff0, ff1, t_0, t_1, data, "2c")
            return self.fine_stage(ff0, ff1, t_0, t_1, data)
        finally:
            _CONV_DTYPE[0] = None


class CasMTR2c(CasMTR4c):
    def __init__(self, config=None, conv_dtype=None):
        super().__init__(config or outdoor_2c_config(), conv_dtype)
