"""hipemu: runs the HIP kernels of croaring_amd/csrc on the host through a wave64 SIMT emulator.
TEST INFRASTRUCTURE ONLY (see shim/hip/hip_runtime.h); the croaring_amd package never uses it."""
from __future__ import annotations

import ctypes as C
import os

from .build_emu import asan_runtime, build  # noqa: F401


def emu_engine():
    """A croaring_amd.Engine object whose C ABI calls land in librhip_emu.so (kernels under hipemu)."""
    from croaring_amd import _lib
    from croaring_amd.engine import Engine

    # HIPEMU_ASAN=1 (with LD_PRELOAD=asan_runtime()): AddressSanitizer build, see build_emu.build
    lib = C.CDLL(build(asan=os.environ.get("HIPEMU_ASAN") == "1"))
    for name, res, args in _lib.SYMBOLS:
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = args

    class EmuEngine(Engine):
        def __init__(self):  # noqa: D401 -- deliberately bypasses _lib.load(): no HIP device involved
            self.lib = lib
            self.h = lib.rhip_ctx_create(-1)
            if not self.h:
                raise RuntimeError("emu ctx_create failed: " + (lib.rhip_last_error() or b"").decode())

        # the emulator's "device" memory is host memory: torch sees it as CPU tensors, and there is no stream
        def torch_device(self):
            import torch
            return torch.device("cpu")

        def torch_stream(self):
            import contextlib
            return contextlib.nullcontext()

        def as_tensor(self, ptr, shape):
            import numpy as np
            import torch
            n = int(np.prod(shape))
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int64)), shape=(n,)).reshape(shape)
            return torch.from_numpy(a)

    return EmuEngine()
