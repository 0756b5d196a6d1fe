"""Drop-in registration: makes the names the reference model imports resolve to this package.

    import casmtr_amd.compat as compat; compat.install()

after which, unchanged reference code such as
    import score_computation_cuda, value_aggregation_cuda                 (functions/quadtree_attention.py:1-2)
    import fast_score_computation                                         (src/model/functions/cascade_functions.py:1)
    from cuda_imp.QuadTreeAttention.QuadtreeAttention.modules.quadtree_attention import QTAttA, QTAttB, ...
                                                                          (src/model/modules/quadtree_attention.py:6)
runs on the HIP kernels.  Signatures follow the pybind modules (score_computation.cpp:35-38, value_aggregation.cpp:62-65,
score_cuda/src/score_computation.cpp:29-32).
"""
import importlib
import sys
import types

from . import ops


def _ext(name, **fns):
    m = types.ModuleType(name)
    m.__doc__ = f"{name}: MI355X implementation provided by casmtr_amd"
    m.__dict__.update(fns)
    return m


def _qta_value_agg_backward(grad_out, score, value, index, grad_score, grad_value):
    ops.qta_value_agg_bwd(grad_out, score, value, index, grad_score, grad_value)


def extension_modules():
    return {
        "score_computation_cuda": _ext(
            "score_computation_cuda",
            score_forward=lambda q, k, i: [ops.qta_score_fwd(q, k, i)],
            score_backward=lambda g, q, k, i: list(ops.qta_score_bwd(g, q, k, i))),
        "value_aggregation_cuda": _ext(
            "value_aggregation_cuda",
            value_aggregation_forward=lambda s, v, i, o: ops.qta_value_agg_fwd(s, v, i, o),
            value_aggregation_backward=_qta_value_agg_backward),
        "fast_score_computation": _ext(
            "fast_score_computation",
            score_forward=lambda q, k, i: [ops.window_score_fwd(q, k, i)],
            score_backward=lambda g, q, k, i: list(ops.window_score_bwd(g, q, k, i))),
    }


def _ensure_parents(full):
    """Parents of `full` must exist for `import a.b.c` to succeed: use the real (reference) packages when they are
    importable, otherwise register empty namespace stand-ins."""
    parts = full.split(".")
    for i in range(1, len(parts)):
        pkg = ".".join(parts[:i])
        if pkg in sys.modules:
            continue
        try:
            importlib.import_module(pkg)
        except ImportError:
            p = types.ModuleType(pkg)
            p.__path__ = []
            sys.modules[pkg] = p
            if i > 1:
                setattr(sys.modules[".".join(parts[: i - 1])], parts[i - 1], p)


def install(packages: bool = True, matching: bool = True, blocks: bool = True):
    """Register the three extension names.  packages=True also aliases the reference's QuadTreeAttention python
    modules (so src/model/modules/quadtree_attention.py:6 picks up the fused QTAttB / CascadeQTAttB); matching=True
    aliases src.model.functions.{coarse_matching,cascade_matching,post_processing} (imported by
    src/model/cascade_model_stage3.py) to the fused matchers; blocks=True aliases src.model.modules.quadtree_attention
    (QuadtreeAttention / CascadeQuadtreeAttention, imported at src/model/modules/transformer.py:12) to the token-major
    callers.  src.model.functions.cascade_functions is left alone: the reference's own file keeps working because it
    only needs `fast_score_computation`."""
    for name, mod in extension_modules().items():
        sys.modules[name] = mod
    aliases = {}
    if packages:
        from .functions import quadtree_attention as fn
        from .modules import quadtree_attention as mods
        aliases["cuda_imp.QuadTreeAttention.QuadtreeAttention.modules.quadtree_attention"] = mods
        aliases["cuda_imp.QuadTreeAttention.QuadtreeAttention.functions.quadtree_attention"] = fn
    if matching:
        from .matching import cascade_matching, coarse_matching, post_processing
        aliases["src.model.functions.coarse_matching"] = coarse_matching
        aliases["src.model.functions.cascade_matching"] = cascade_matching
        aliases["src.model.functions.post_processing"] = post_processing
    if blocks:
        from .modules import quadtree_block
        aliases["src.model.modules.quadtree_attention"] = quadtree_block
    for full, mod in aliases.items():   # leaves first: a parent package imported below must already see them
        sys.modules[full] = mod
    for full, mod in aliases.items():
        _ensure_parents(full)
        parent = sys.modules[full.rsplit(".", 1)[0]]
        setattr(parent, full.rsplit(".", 1)[1], mod)
