#!/usr/bin/env python3
"""Builds the REFERENCE's whole CasMTR-4c model (its own, unmodified source under /root/reference) in a fresh interpreter and
prints a JSON description of it: parameter names/shapes and the defining module of every hot-path class instance.

    python tests/dropin_probe.py reference     hot path = the reference's python modules (extensions stubbed out)
    python tests/dropin_probe.py casmtr_amd    after `casmtr_amd.compat.install()`: hot path = this package

tests/test_dropin_reference_model.py runs both and compares.  Build container only (needs /root/reference)."""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "golden"))

import torch  # noqa: E402

import ref_stubs  # noqa: E402


def main(mode):
    if mode == "casmtr_amd":
        import casmtr_amd.compat as compat
        compat.install()                      # BEFORE any `src.model...` import, as INTEGRATION.md says
    else:
        for n in ("score_computation_cuda", "value_aggregation_cuda", "fast_score_computation"):
            ref_stubs.stub(n)
    ref_stubs.install_third_party()
    ref_stubs.install_full_model_extras()
    from configs.default import get_cfg_defaults
    cfg = get_cfg_defaults()
    cfg.merge_from_file(os.path.join(REF, "configs/model_configs/outdoor/loftr_ds_quadtree_cas_twins_large_stage3.py"))
    mc = ref_stubs.lower(cfg)["loftr"]
    mc["coarse2"]["post_config"]["method"] = "maxpool_nms"
    mc["coarse2"]["post_config"]["window_size"] = 5
    from src.model.cascade_model_stage3 import CasMTR
    torch.manual_seed(0)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        model = CasMTR(config=mc).eval()
    hot = ("QTAttB", "CascadeQTAttB", "QuadtreeAttention", "CascadeQuadtreeAttention", "CoarseMatching", "CascadeMatching")
    classes = {}
    for name, m in model.named_modules():
        cn = type(m).__name__
        if cn in hot:
            classes.setdefault(cn, {"module": type(m).__module__, "count": 0})["count"] += 1
    out = {"params": {k: list(v.shape) for k, v in model.state_dict().items()}, "classes": classes}
    print("PROBE_JSON " + json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1])
