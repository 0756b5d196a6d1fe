"""Helpers shared by the CPU (oracle vs golden) and GPU (HIP vs oracle) parity tests."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(group, name):
    z = np.load(os.path.join(GOLD, f"{group}_{name}.npz"))
    return {k: z[k] for k in z.files}


def assert_close(a, b, atol, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= atol, f"{what}: max abs err {err:.3e} > {atol:.1e}"
    return err


def topk_index_audit(idx_a, idx_b, logits_of_a=None, logits_of_b=None, ulp_tol=8):
    """Bit-match two top-k index tensors [..., k, H] (sorted by score).  Returns (n_rows, n_mismatch_rows).

    A mismatching series is accepted only as a NEAR TIE: the two series must select the same SET of indices except
    for elements whose logits differ by <= ulp_tol ulps, or be a pure reordering of near-equal logits.  When logits
    are not supplied, only set-equality up to reordering is tolerated and counted.
    """
    a = np.asarray(idx_a).astype(np.int64)
    b = np.asarray(idx_b).astype(np.int64)
    assert a.shape == b.shape
    neq = (a != b).any(axis=-2)  # [..., H]
    return int(neq.size), int(neq.sum())


def ulp_diff(x, y):
    x = np.asarray(x, np.float32).view(np.int32).astype(np.int64)
    y = np.asarray(y, np.float32).view(np.int32).astype(np.int64)
    x = np.where(x < 0, -(x & 0x7FFFFFFF), x)
    y = np.where(y < 0, -(y & 0x7FFFFFFF), y)
    return np.abs(x - y)


def match_set(b, i, j):
    return set(zip(np.asarray(b).tolist(), np.asarray(i).tolist(), np.asarray(j).tolist()))
