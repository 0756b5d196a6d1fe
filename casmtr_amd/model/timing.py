"""Whole-model timing of CasMTR4c: random-init weights, synthetic image pairs, per-stage HIP-event breakdown.
A SECOND metric next to bench.py's hot-path line -- it includes the torch-op glue (backbone convolutions, MLP GEMMs, window
self-attention) that the hot-path step leaves out.  Used by bench.py ('whole_model' object) and tools/model_e2e_time.py."""
import torch

from ..modules.quadtree_block import set_caller_layout
from .casmtr4c import CasMTR4c, outdoor_2c_config, outdoor_4c_config


def time_whole_model(batch=8, size=832, steps=5, warmup=2, coarse_thr=None, cascade_thr=None, device="cuda", model="4c", conv_dtype=None,
                     attn_layout="quads", proj_gemm="split"):
    """attn_layout: the attention blocks' route (modules/quadtree_block.py::_quad_route); the timing opts into the quad-major kernels
    of the hot path, the modules' own default is token-major"""
    if model == "indoor":
        return _time_indoor(batch, steps, warmup, conv_dtype, device, attn_layout, proj_gemm)
    cfg = outdoor_2c_config() if model == "2c" else outdoor_4c_config()
    if coarse_thr is not None:
        cfg["match_coarse"]["thr"] = coarse_thr
    if cascade_thr is not None:
        cfg["match_cascade"].update(test_thr=cascade_thr, pre_thr=[0.0])
        if "match_cascade_2c" in cfg:
            cfg["match_cascade_2c"].update(test_thr=cascade_thr, pre_thr=[0.0, 0.0])
    torch.manual_seed(0)
    m = set_caller_layout(CasMTR4c(cfg, conv_dtype=conv_dtype).eval().to(device), attn_layout, proj_gemm)
    g = torch.Generator(device=device).manual_seed(1)
    mk = lambda: torch.rand((batch, 3, size, size), device=device, generator=g)
    sets = [(mk(), mk()) for _ in range(2)]
    names = ["backbone", "stage_8c", "stage_4c"] + (["stage_2c"] if model == "2c" else []) + ["fine"]
    acc = dict.fromkeys(names, 0.0)
    nm = 0

    from . import casmtr4c as _mod

    def step(i, timed):
        nonlocal nm
        _mod._CONV_DTYPE[0] = conv_dtype   # the staged calls below bypass forward(), which normally sets it
        im0, im1 = sets[i % 2]
        data = {"image0": im0, "image1": im1}
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        with torch.no_grad():
            ev[0].record()
            (f8_0, f8_1), (f4_0, f4_1), (ff0, ff1) = m.features(data)
            ev[1].record()
            t8 = m.coarse_stage(f8_0, f8_1, data)
            ev[2].record()
            t = m.cascade_stage(f4_0, f4_1, *t8, data, "4c")
            ev[3].record()
            if model == "2c":
                t = m.cascade_stage(ff0, ff1, *t, data, "2c")
                ev[4].record()
            m.fine_stage(ff0, ff1, *t, data)
            ev[-1].record()
        torch.cuda.synchronize()
        if timed:
            for k, n in enumerate(names):
                acc[n] += ev[k].elapsed_time(ev[k + 1])
            nm += int(data["mkpts0_f"].shape[0])

    for i in range(warmup):
        step(i, False)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for i in range(steps):
        step(i, True)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / steps
    out = {"metric": f"whole-model image pairs/sec (CasMTR-{model}: torch glue + HIP hot path; attention / matching fp32)", "value": round(batch / ms * 1e3, 2),
           "unit": "pairs/s", "ms_per_step": round(ms, 2), "batch": batch, "size": size, "steps": steps,
           "stage_ms": {k: round(v / steps, 2) for k, v in acc.items()}, "matches_per_pair": round(nm / steps / batch, 1),
           "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "data": "synthetic", "weights": "random-init"}
    _mod._CONV_DTYPE[0] = None
    out["conv_dtype"] = "fp32" if conv_dtype is None else str(conv_dtype).replace("torch.", "")
    out["attention_layout"] = attn_layout
    out["attention_projection_gemm"] = proj_gemm
    del m, sets
    torch.cuda.empty_cache()
    return out


def _time_indoor(batch, steps, warmup, conv_dtype, device, attn_layout="quads", proj_gemm="split"):
    """CasMTRIndoor4c on 640x480 frames (BASELINE configs[4] shapes)"""
    from . import casmtr4c as _mod
    from .indoor import CasMTRIndoor4c
    torch.manual_seed(0)
    m = set_caller_layout(CasMTRIndoor4c(conv_dtype=conv_dtype).eval().to(device), attn_layout, proj_gemm)
    g = torch.Generator(device=device).manual_seed(1)
    mk = lambda: torch.rand((batch, 3, 480, 640), device=device, generator=g)
    sets = [(mk(), mk()) for _ in range(2)]
    names = ["backbone", "stage_8c", "stage_4c", "fine"]
    acc = dict.fromkeys(names, 0.0)
    nm = 0

    def step(i, timed):
        nonlocal nm
        _mod._CONV_DTYPE[0] = conv_dtype
        data = {"image0": sets[i % 2][0], "image1": sets[i % 2][1]}
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        with torch.no_grad():
            ev[0].record()
            x, f8, f4, ff = m.features(data)
            ev[1].record()
            t8 = m.coarse_stage(f8, data)
            ev[2].record()
            t4_0, t4_1, ff0, ff1 = m.cascade_stage(x, f4, ff, *t8, data)
            ev[3].record()
            m.fine_stage(ff0, ff1, t4_0, t4_1, data)
            ev[4].record()
        torch.cuda.synchronize()
        if timed:
            for k, n in enumerate(names):
                acc[n] += ev[k].elapsed_time(ev[k + 1])
            nm += int(data["mkpts0_f"].shape[0])

    for i in range(warmup):
        step(i, False)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for i in range(steps):
        step(i, True)
    t1.record()
    torch.cuda.synchronize()
    _mod._CONV_DTYPE[0] = None
    ms = t0.elapsed_time(t1) / steps
    out = {"metric": "whole-model image pairs/sec (CasMTR indoor 4c, 640x480: torch glue + HIP hot path; attention / matching fp32)",
           "value": round(batch / ms * 1e3, 2), "unit": "pairs/s", "ms_per_step": round(ms, 2), "batch": batch, "size": "640x480",
           "steps": steps, "stage_ms": {k: round(v / steps, 2) for k, v in acc.items()}, "matches_per_pair": round(nm / steps / batch, 1),
           "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "data": "synthetic", "weights": "random-init",
           "conv_dtype": "fp32" if conv_dtype is None else str(conv_dtype).replace("torch.", ""), "attention_layout": attn_layout, "attention_projection_gemm": proj_gemm}
    del m, sets
    torch.cuda.empty_cache()
    return out
