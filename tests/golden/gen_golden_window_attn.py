#!/usr/bin/env python3
"""Fixture for the padded-window quirk of the reference's GroupAttention.forward_mask (src/model/modules/cascade_attention.py:128-157):
`mask[:, -pad_b:, :].fill_(1)` with pad_b == 0 selects the WHOLE mask, so whenever exactly one grid side is a multiple of the
window size the mask becomes all ones, attn_mask is 0 everywhere and real queries also attend to the zero-padded keys (whose k / v
are the qkv bias).  Stored: outputs of the reference module for grids with (pad_b, pad_r) = (0, >0), (>0, 0), (>0, >0) and (0, 0),
with deterministic weights; inputs are regenerated from the seed by the test.

    python tests/golden/gen_golden_window_attn.py        (build container only; needs /root/reference)

Nothing of the reference is copied: the module is imported and called, only numbers are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import gen_golden as gg  # noqa: E402,F401  installs the extension / kornia / timm stubs and puts /root/reference on sys.path
import ref_stubs  # noqa: E402

from golden_inputs import (WINDOW_ATTN_DIM as DIM, WINDOW_ATTN_GRIDS as GRIDS, WINDOW_ATTN_HEADS as HEADS, WINDOW_ATTN_WS as WS,  # noqa: E402
                           window_attn_tokens as tokens, window_attn_weights as weights)


def main():
    ref_stubs.install_full_model_extras()
    from src.model.modules.cascade_attention import GroupAttention
    m = GroupAttention(DIM, num_heads=HEADS, qkv_bias=True, ws=WS).eval()
    m.load_state_dict(weights())
    out = {}
    with torch.no_grad():
        for H, W in GRIDS:
            out[f"out_{H}x{W}"] = m(tokens(H, W), H, W).numpy()
    np.savez_compressed(os.path.join(HERE, "window_attn_padding.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
