"""CPU, build container only: the zero-edit route of INTEGRATION.md.  The reference's whole CasMTR-4c model is constructed
from its own unmodified source twice, in fresh interpreters -- once on its python modules, once after
`casmtr_amd.compat.install()` -- and must come out with the same state-dict (names and shapes), with every hot-path
module instance (QTAttB, CascadeQTAttB, their callers, both matchers) provided by this package."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/src/model"), reason="reference checkout not present")


def _probe(mode):
    r = subprocess.run([sys.executable, os.path.join(HERE, "dropin_probe.py"), mode], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("PROBE_JSON ")][-1]
    return json.loads(line[len("PROBE_JSON "):])


def test_reference_model_builds_on_casmtr_amd_with_identical_state_dict():
    ref, ours = _probe("reference"), _probe("casmtr_amd")
    assert ref["params"] == ours["params"], "state-dict names / shapes differ: checkpoints would not load"
    assert len(ref["params"]) > 300
    for cls, info in ref["classes"].items():
        assert not info["module"].startswith("casmtr_amd")
        assert cls in ours["classes"] and ours["classes"][cls]["count"] == info["count"], cls
    for cls, info in ours["classes"].items():
        assert info["module"].startswith("casmtr_amd."), f"{cls} still comes from {info['module']}"
    # the 4c outdoor model: 6 coarse QuadTree layers, 2 cascade cross layers, one matcher per stage
    assert ours["classes"]["QTAttB"]["count"] == 6 and ours["classes"]["CascadeQTAttB"]["count"] == 2
    assert ours["classes"]["CoarseMatching"]["count"] == 1 and ours["classes"]["CascadeMatching"]["count"] == 1
