"""Child process of test_pinned_allocator_hook (tests/test_gpu_compat.py, tests/test_emu_kernels.py): installs the
library's pinned allocator as the REFERENCE's memory hook (roaring_init_memory_hook, include/roaring/memory.h:29-38), then
builds bitmaps with the reference, runs the drop-in set operations on them and lets the reference free everything.
argv[1] = "hip" (libroaring_hip.so on a GPU) or "emu" (the CPU emulator build).  A process of its own: the hook is
process-wide and permanent."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.pyoracle import Ref  # noqa: E402
from gen_inputs import random_bitmap  # noqa: E402

# the reference's symbols must be visible to dlsym(RTLD_DEFAULT, ...) inside the engine library
C.CDLL(Ref.PATH, mode=C.RTLD_GLOBAL)
ref = Ref()
if sys.argv[1] == "emu":
    from emu import build_emu
    lib = C.CDLL(build_emu.build())
else:
    import croaring_amd
    lib = croaring_amd.load()
vp = C.c_void_p
for op in ("and", "or", "xor", "andnot"):
    f = getattr(lib, f"roaring_bitmap_{op}"); f.restype = vp; f.argtypes = [vp, vp]
    f = getattr(lib, f"roaring_bitmap_{op}_inplace"); f.restype = None; f.argtypes = [vp, vp]
if len(sys.argv) > 2 and sys.argv[2] == "allocfail":
    # test_inplace_dropins_survive_transient_alloc_failure: a device allocation failure that PERSISTS inside a void
    # in-place drop-in ends the process (abort) with the reason on stderr
    lib.rhip_debug_fail_allocs.restype = None
    lib.rhip_debug_fail_allocs.argtypes = [C.c_int, C.c_int]
    rng = np.random.default_rng(5)
    a, b = ref.from_sorted(random_bitmap(rng)), ref.from_sorted(random_bitmap(rng))
    getattr(lib, "roaring_bitmap_or_inplace")(a, b)  # (a working call first: the lane exists)
    keys = np.sort(rng.choice(4000, 900, replace=False)).astype(np.uint32)  # (larger than what the lane has recycled: it must allocate)
    big = np.concatenate([(k << np.uint32(16)) | np.unique(rng.integers(0, 65536, 500)).astype(np.uint32) for k in keys])
    a, b = ref.from_sorted(big), ref.from_sorted(big[::2].copy())
    lib.rhip_debug_fail_allocs(0, 1 << 30)
    getattr(lib, "roaring_bitmap_xor_inplace")(a, b)
    print("not reached: the call above must abort")
    sys.exit(0)
lib.rhip_install_pinned_allocator.restype = C.c_int
lib.rhip_install_pinned_allocator.argtypes = [C.c_size_t]
lib.rhip_pinned_allocator_stats.restype = C.c_int
rc = lib.rhip_install_pinned_allocator(8 << 20)  # a small arena: the fall-through to the C library is exercised too
assert rc == 0, ("install failed", rc, lib.rhip_last_error())
st = (C.c_ulonglong * 4)()
ref.L.roaring_bitmap_copy.restype = vp
ref.L.roaring_bitmap_copy.argtypes = [vp]
rng = np.random.default_rng(77)
live = []
for it in range(30):
    a, b = ref.from_sorted(random_bitmap(rng)), ref.from_sorted(random_bitmap(rng))  # allocated through the hook
    for op in ("and", "or", "xor", "andnot"):
        want = ref.op(op, a, b)
        got = getattr(lib, f"roaring_bitmap_{op}")(a, b)
        assert got and ref.validate(got) and ref.serialize(got) == ref.serialize(want), (it, op)
        a2 = ref.L.roaring_bitmap_copy(a)
        getattr(lib, f"roaring_bitmap_{op}_inplace")(a2, b)  # frees a2's old containers (hook memory) and installs ours
        assert ref.validate(a2) and ref.serialize(a2) == ref.serialize(want), (it, op, "inplace")
        for h in (want, got, a2):
            ref.free(h)  # the REFERENCE frees what the drop-in allocated
    live += [a, b]
    if it == 10:
        lib.rhip_pinned_allocator_stats(st)
        assert st[2] > 1000, list(st)  # blocks were served from the pinned arena
for h in live:
    ref.free(h)
lib.rhip_pinned_allocator_stats(st)
print("pinned allocator: arena %d B, in use at exit %d B, served %d, passed to libc %d" % tuple(st))
assert st[1] == 0, "blocks of the pinned arena leaked"
assert st[2] > 1000 and st[3] > 0
print("hook ok")
