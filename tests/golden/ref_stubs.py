"""Stand-ins for the third-party packages the reference imports but this image lacks (kornia, timm, yacs, cv2, h5py, loguru):
only the handful of symbols its model files touch.  Used by the fixture generators and by tests/dropin_probe.py; none of
this is reference code, and none of it is used by the product."""
import copy
import importlib.util
import sys
import types

import torch
import torch.nn as nn


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_third_party():
    """kornia.{feature,utils.grid.create_meshgrid} and timm.models.layers.{DropPath,to_2tuple,trunc_normal_}."""
    k = stub("kornia")
    kf = stub("kornia.feature")
    kf.__all__ = []
    k.feature = kf
    ku = stub("kornia.utils")
    kug = stub("kornia.utils.grid")

    def create_meshgrid(h, w, normalized_coordinates=True, device=None, dtype=torch.float32):
        if normalized_coordinates:   # kornia: linspace(-1, 1) along each axis
            ys, xs = torch.linspace(-1, 1, h, device=device, dtype=dtype), torch.linspace(-1, 1, w, device=device, dtype=dtype)
        else:
            ys, xs = torch.arange(h, device=device, dtype=dtype), torch.arange(w, device=device, dtype=dtype)
        ys, xs = torch.meshgrid(ys, xs, indexing="ij")
        return torch.stack([xs, ys], -1)[None]

    kug.create_meshgrid = create_meshgrid
    ku.grid = kug
    ku.create_meshgrid = create_meshgrid
    k.utils = ku
    stub("timm")
    stub("timm.models")
    tl = stub("timm.models.layers")

    class DropPath(nn.Identity):
        def __init__(self, *a, **k):
            super().__init__()

    tl.DropPath = DropPath
    tl.to_2tuple = lambda x: (x, x) if not isinstance(x, (tuple, list)) else tuple(x)
    tl.trunc_normal_ = lambda t, std=1.0, **k: nn.init.trunc_normal_(t, std=std)


class CN(dict):
    """the sliver of yacs.config.CfgNode that configs/default.py and the model config files use"""
    def __init__(self, d=None):
        super().__init__()

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def merge_from_file(self, path):
        spec = importlib.util.spec_from_file_location("cfgfile", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)

        def upd(a, b):
            for k, v in b.items():
                if isinstance(v, CN) and isinstance(a.get(k), CN):
                    upd(a[k], v)
                else:
                    a[k] = copy.deepcopy(v)
        upd(self, mod.cfg)


def install_full_model_extras():
    """what importing the WHOLE model additionally needs: yacs, kornia.geometry.subpix.dsnt, cv2, h5py, loguru."""
    y = stub("yacs")
    y.config = stub("yacs.config", CfgNode=CN)
    k = sys.modules["kornia"]
    kg, kgs, kgd = stub("kornia.geometry"), stub("kornia.geometry.subpix"), stub("kornia.geometry.subpix.dsnt")

    def spatial_expectation2d(inp, normalized_coordinates=True):
        b, c, h, w = inp.shape
        if normalized_coordinates:
            xs, ys = torch.linspace(-1, 1, w), torch.linspace(-1, 1, h)
        else:
            xs, ys = torch.arange(w).float(), torch.arange(h).float()
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        x = inp.view(b, c, -1)
        return torch.stack([(x * gx.reshape(-1)).sum(-1), (x * gy.reshape(-1)).sum(-1)], -1)

    kgd.spatial_expectation2d = spatial_expectation2d
    kgs.dsnt, kg.subpix, k.geometry = kgd, kgs, kg
    for n in ("cv2", "h5py"):
        stub(n)
    stub("loguru", logger=types.SimpleNamespace(info=print, warning=print, error=print, debug=print))


def lower(c):
    return {k.lower(): lower(v) for k, v in c.items()} if isinstance(c, CN) else c
