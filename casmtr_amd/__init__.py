"""casmtr_amd -- CasMTR's cascaded-matching hot path, written from scratch for MI355X (gfx950).

Layers (bottom up):
  csrc/*.hip + include/casmtr_hip.h   hand-written HIP kernels behind a C ABI (libcasmtr_hip.so)
  _lib.py / ops.py                    ctypes binding + tensor-level front end (stream / device plumbing only)
  functions/, modules/, matching/     the reference's autograd.Function / nn.Module surface, same names & semantics
  compat.py                           registers the reference's extension-module and package names (drop-in)
  pipeline.py / dist.py               the hot-path chain used by bench.py and its one-process-per-GPU sharding

There is no CPU fallback anywhere in this package.
"""
__version__ = "0.1.0"
