"""ctypes binding of libcasmtr_hip.so (the C ABI declared in include/casmtr_hip.h).

There is NO CPU fallback: if the library is missing or a kernel reports an error, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CASMTR_LIB_PATH") or os.path.join(_HERE, "lib", "libcasmtr_hip.so")   # override: A/B of build variants (tools/)
CSRC = os.path.join(_HERE, "csrc")
ERR_UNSUPPORTED = 1001

_P, _I, _F, _SZ, _LL = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_longlong

# name -> (restype, argtypes): exactly the prototypes of include/casmtr_hip.h
SIGNATURES = {
    "casmtr_abi_version": (_I, []),
    "casmtr_debug_work_counters_nonzero": (_I, []),
    "casmtr_qta_score_fwd": (_I, [_P] * 4 + [_I] * 6 + [_P]),
    "casmtr_qta_score_bwd": (_I, [_P] * 6 + [_I] * 6 + [_P]),
    "casmtr_qta_value_agg_fwd": (_I, [_P] * 4 + [_I] * 6 + [_P]),
    "casmtr_qta_value_agg_bwd": (_I, [_P] * 6 + [_I] * 6 + [_P]),
    "casmtr_window_score_fwd": (_I, [_P] * 4 + [_I] * 5 + [_P]),
    "casmtr_window_score_bwd": (_I, [_P] * 6 + [_I] * 5 + [_P]),
    "casmtr_nchw_to_tokens": (_I, [_P, _P, _I, _I, _I, _P]),
    "casmtr_nchw_to_tokens_multi": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "casmtr_qta_coarse_level_fwd": (_I, [_P, _P, _P, _F, _I, _F, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "casmtr_qta_coarse_level_ws_floats": (_SZ, [_I] * 4),
    "casmtr_qta_coarse_level_ws_floats_k": (_SZ, [_I] * 5),
    "casmtr_qta_fine_level_fwd": (_I, [_P, _P, _P, _P, _F, _I, _F, _P, _P, _P, _P, _P] + [_I] * 8 + [_P]),
    "casmtr_nchw_to_quads_multi": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "casmtr_tokens_to_quads": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "casmtr_topk_idx_to_tab": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "casmtr_qta_fine_level_quad_fwd": (_I, [_P, _P, _P, _P, _F, _I, _F, _P, _P, _P, _P, _P, _P] + [_I] * 8 + [_P]),
    "casmtr_qta_coarse_level_tab_fwd": (_I, [_P, _P, _P, _F, _I, _F, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "casmtr_cascade_attn_quad_fwd": (_I, [_P, _P, _P, _P, _P, _F, _P] + [_I] * 8 + [_P]),
    "casmtr_cascade_attn_fwd": (_I, [_P, _P, _P, _P, _P, _F, _I, _P, _P] + [_I] * 8 + [_P]),
    "casmtr_window_warp_idx": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "casmtr_dual_softmax_fwd": (_I, [_P, _P, _P, _P, _F, _I, _F, _I, _P, _I, _I, _I, _I, _I, _P, _P,
                                     _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "casmtr_dual_softmax_ws_bytes": (_SZ, [_I] * 3),
    "casmtr_dual_softmax_split_fwd": (_I, [_P, _P, _P, _P, _F, _I, _F, _I, _P, _I, _I, _I, _I, _I, _P, _P,
                                           _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "casmtr_dual_softmax_split_ws_bytes": (_SZ, [_I] * 4),
    "casmtr_window_match_fwd": (_I, [_P, _P, _P, _P, _P, _F, _I, _P, _P, _P] + [_I] * 7 + [_P]),
    "casmtr_window_match_pos_fwd": (_I, [_P, _P, _P, _P, _P, _F, _I, _I, _P, _P, _P] + [_I] * 7 + [_P]),
    "casmtr_window_expand_idx": (_I, [_P, _P] + [_I] * 7 + [_P]),
    "casmtr_nms_select_fwd": (_I, [_P, _P, _P, _I, _F, _P, _I, _I, _F, _P, _I, _I, _F, _I, _P, _I, _P,
                                   _P, _P, _P, _P, _P] + [_I] * 5 + [_P, _P]),
    "casmtr_nms_select_ws_bytes": (_SZ, [_I] * 3),
    "casmtr_linear_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "casmtr_token_pool_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "casmtr_linear_quads_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "casmtr_quad_pool_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "casmtr_linear_split_prep_bytes": (_SZ, [_I, _I]),
    "casmtr_linear_split_prep": (_I, [_P, _P, _I, _I, _P]),
    "casmtr_linear_split_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "casmtr_linear_split_pyramid_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "casmtr_dwconv3x3_tokens_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "casmtr_layer_norm_fwd": (_I, [_P, _P, _P, _P, _P, _LL, _I, _F, _P]),
    "casmtr_window_attn_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    "casmtr_pola_attn_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    "casmtr_prof_enable": (None, [_I]),
    "casmtr_debug_set": (None, [_I]),
    "casmtr_prof_enable_only": (_I, [_I]),
    "casmtr_prof_reserve": (_I, [_I]),
    "casmtr_prof_read": (_I, [_I, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "casmtr_prof_name": (C.c_char_p, [_I]),
    "casmtr_prof_symbol": (C.c_char_p, [_I]),
    "casmtr_prof_read_all": (_I, [_I, C.POINTER(C.c_double), _I]),
}
PROF_COUNT = 21


def prof_enable(on: bool):
    lib().casmtr_prof_enable(int(on))


def prof_enable_only(name: str):
    """Time only the kernel called `name` (as reported by prof_read)."""
    ids = {lib().casmtr_prof_name(i).decode(): i for i in range(PROF_COUNT)}
    check(lib().casmtr_prof_enable_only(ids[name]), "prof_enable_only")


def prof_reserve(pairs: int):
    """create the event pairs a timed region will need up front"""
    check(lib().casmtr_prof_reserve(int(pairs)), "prof_reserve")


def prof_symbols():
    """-> {scope name: kernel symbol(s) that ran under it since it was last timed}"""
    out = {}
    for i in range(PROF_COUNT):
        s = lib().casmtr_prof_symbol(i).decode()
        if s:
            out[lib().casmtr_prof_name(i).decode()] = s
    return out


def prof_read_all(name: str):
    """-> the individual launch durations (ms) of scope `name`, in launch order"""
    ids = {lib().casmtr_prof_name(i).decode(): i for i in range(PROF_COUNT)}
    n = lib().casmtr_prof_read_all(ids[name], None, 0)
    if n <= 0:
        return []
    buf = (C.c_double * n)()
    if lib().casmtr_prof_read_all(ids[name], buf, n) != n:
        raise RuntimeError("prof_read_all")
    return list(buf)


def prof_read():
    """-> {scope name: (total_ms, launches)} for every scope timed since prof_enable(True)."""
    out = {}
    for i in range(PROF_COUNT):
        ms, n = C.c_double(0.0), C.c_int(0)
        check(lib().casmtr_prof_read(i, C.byref(ms), C.byref(n)), "prof_read")
        if n.value:
            out[lib().casmtr_prof_name(i).decode()] = (ms.value, n.value)
    return out

_lib = None


def build(force: bool = False) -> str:
    """Compile the HIP sources for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-j4"] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(casmtr_amd has no CPU fallback)")
        # In a PyTorch process the HIP runtime must be the one torch ships (torch/lib/libamdhip64.so): loading this library first would
        # bind the process to /opt/rocm's copy of the same SONAME, and torch's streams / allocations then belong to a runtime this
        # library does not see (every launch fails with hipErrorNoDevice).  The library itself has no torch dependency.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the ABI is incomplete
            fn.restype, fn.argtypes = res, args
        if l.casmtr_abi_version() != 8:
            raise RuntimeError("libcasmtr_hip.so ABI version mismatch")
        _lib = l
    return _lib


def check(code: int, what: str):
    if code == 0:
        return
    if code == ERR_UNSUPPORTED:
        raise RuntimeError(f"{what}: shape not supported by the HIP kernel (CASMTR_ERR_UNSUPPORTED)")
    raise RuntimeError(f"{what}: HIP error {code}")
