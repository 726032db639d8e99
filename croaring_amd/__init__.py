"""croaring_amd -- MI355X-native Roaring-bitmap set-operation engine (host-side mirror).

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C ABI of
libroaring_hip.so, declared in include/roaring_hip.h) and this thin ctypes mirror of the
reference's operator interface.  Importing the package does not need a GPU; creating an
`Engine` does, and fails loudly without one.
"""
from ._lib import LIB_PATH, RoaringHipError, load  # noqa: F401
from .engine import Batch, Engine, Pool, OPS, synth_sparse_portable  # noqa: F401

__all__ = ["Batch", "Engine", "Pool", "OPS", "synth_sparse_portable", "RoaringHipError", "load", "LIB_PATH"]
