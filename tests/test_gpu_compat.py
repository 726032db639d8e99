"""GPU test of the CRoaring-named drop-in entry points (include/roaring_hip_compat.h): operands are
REAL reference structs (built by oracle/_ref), results come back as reference-layout structs that the
reference's own serialize / validate / free accept -- struct-layout compatibility in both directions."""
import ctypes as C

import numpy as np
import pytest

from gen_inputs import random_bitmap
from util import OPS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import croaring_amd
    return bind(croaring_amd.load())


def bind(lib):
    """ctypes prototypes of the CRoaring-named entry points (shared with tests/test_emu_kernels.py)."""
    vp = C.c_void_p
    for op in OPS:
        f = getattr(lib, f"roaring_bitmap_{op}"); f.restype = vp; f.argtypes = [vp, vp]
        f = getattr(lib, f"roaring_bitmap_{op}_inplace"); f.restype = None; f.argtypes = [vp, vp]
        f = getattr(lib, f"roaring_bitmap_{op}_cardinality"); f.restype = C.c_uint64; f.argtypes = [vp, vp]
    lib.roaring_bitmap_or_many.restype = vp; lib.roaring_bitmap_or_many.argtypes = [C.c_size_t, C.POINTER(vp)]
    lib.roaring_bitmap_xor_many.restype = vp; lib.roaring_bitmap_xor_many.argtypes = [C.c_size_t, C.POINTER(vp)]
    lib.roaring_bitmap_or_many_heap.restype = vp
    lib.roaring_bitmap_or_many_heap.argtypes = [C.c_uint32, C.POINTER(vp)]
    return lib


def set_cow(handle, on=True):
    # roaring_array_t: int32 size, int32 allocation_size, 3 pointers, uint8 flags (roaring_types.h:61-68)
    flags = C.cast(handle + 32, C.POINTER(C.c_uint8))
    flags[0] = (flags[0] | 1) if on else (flags[0] & ~1)


def get_flags(handle):
    return C.cast(handle + 32, C.POINTER(C.c_uint8))[0]


def test_dropin_pairwise(hip, ref):
    rng = np.random.default_rng(5)
    ref.L.roaring_bitmap_copy.restype = C.c_void_p
    ref.L.roaring_bitmap_copy.argtypes = [C.c_void_p]
    for it in range(25):
        a, b = ref.from_sorted(random_bitmap(rng)), ref.from_sorted(random_bitmap(rng))
        cow = it % 3 == 0
        keep = None
        if cow:
            set_cow(a); set_cow(b)
            keep = ref.L.roaring_bitmap_copy(a)  # a's containers become shared (refcounted) wrappers
        for op in OPS:
            want = ref.op(op, a, b)
            got = getattr(hip, f"roaring_bitmap_{op}")(a, b)
            assert got, f"drop-in roaring_bitmap_{op} returned NULL (no device?)"
            assert got, "drop-in returned NULL"
            assert ref.validate(got)
            assert ref.serialize(got) == ref.serialize(want), (it, op)
            assert (get_flags(got) & 1) == (1 if cow else 0)
            assert getattr(hip, f"roaring_bitmap_{op}_cardinality")(a, b) == ref.cardinality(want)
            # in-place form on a copy of a
            a2 = ref.L.roaring_bitmap_copy(a)
            getattr(hip, f"roaring_bitmap_{op}_inplace")(a2, b)
            assert ref.validate(a2)
            assert np.array_equal(ref.to_array(a2), ref.to_array(want)), (it, op, "inplace")
            assert ref.serialize(a2) == ref.serialize(want)
            for h in (want, got, a2):
                ref.free(h)  # the REFERENCE's roaring_bitmap_free releases our allocations
        if keep:
            ref.free(keep)
        ref.free(a)
        ref.free(b)


def test_dropin_reentrant(hip, ref):
    """The reference's set operations may be called from any number of threads on distinct bitmaps
    (roaring.h:102-113); so may the drop-ins: every call takes a lane -- a context of its own -- for its duration
    (roaring_compat.inc).  Eight threads issue their own pairwise / in-place / cardinality / many-way calls side by side
    (ctypes releases the GIL around each) and every result is checked against the reference."""
    import threading
    ref.L.roaring_bitmap_copy.restype = C.c_void_p
    ref.L.roaring_bitmap_copy.argtypes = [C.c_void_p]
    n_threads, iters = 8, 12
    rngs = [np.random.default_rng(100 + t) for t in range(n_threads)]
    inputs = [[(ref.from_sorted(random_bitmap(rngs[t])), ref.from_sorted(random_bitmap(rngs[t]))) for _ in range(iters)]
              for t in range(n_threads)]
    wants = [[{op: ref.op(op, a, b) for op in OPS} for a, b in inputs[t]] for t in range(n_threads)]
    got = [[{} for _ in range(iters)] for _ in range(n_threads)]
    start = threading.Barrier(n_threads)

    def work(t):
        start.wait()
        for i, (a, b) in enumerate(inputs[t]):
            for op in OPS:
                r = getattr(hip, f"roaring_bitmap_{op}")(a, b)
                card = getattr(hip, f"roaring_bitmap_{op}_cardinality")(a, b)
                a2 = ref.L.roaring_bitmap_copy(a)
                getattr(hip, f"roaring_bitmap_{op}_inplace")(a2, b)
                arr = (C.c_void_p * 2)(a, b)
                m = hip.roaring_bitmap_or_many(2, arr) if op == "or" else None
                got[t][i][op] = (r, card, a2, m)

    ths = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    for t in range(n_threads):
        for i in range(iters):
            for op in OPS:
                r, card, a2, m = got[t][i][op]
                want = wants[t][i][op]
                assert r, "drop-in returned NULL (no device?)"
                ws = ref.serialize(want)
                assert ref.validate(r) and ref.serialize(r) == ws, (t, i, op)
                assert card == ref.cardinality(want), (t, i, op)
                assert ref.validate(a2) and ref.serialize(a2) == ws, (t, i, op, "inplace")
                if m is not None:
                    assert ref.validate(m) and np.array_equal(ref.to_array(m), ref.to_array(want)), (t, i, "or_many")
                    ref.free(m)
                for h in (r, a2, want):
                    ref.free(h)
            a, b = inputs[t][i]
            ref.free(a)
            ref.free(b)


def test_pinned_allocator_hook(ref):
    """rhip_install_pinned_allocator: the library's page-locked arena installed as the reference's memory hook; bitmaps
    built by the reference, operated on by the drop-ins, freed by the reference (tests/hook_child.py, a process of its
    own: the hook is process-wide)."""
    import os, subprocess, sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hook_child.py")
    p = subprocess.run([sys.executable, child, "hip"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "hook ok" in p.stdout, (p.stdout + p.stderr)[-3000:]


def dropin_many_body(hip, ref, iters=24):
    """The many-way drop-ins against the functions they replace, BYTES: roaring_bitmap_or_many and roaring_bitmap_xor_many
    (fixed folds, replayed on the device) and roaring_bitmap_or_many_heap (the size-ordered tournament, replayed step by
    step: rhip_or_many_heap)."""
    from gen_inputs import random_bitmap
    rng = np.random.default_rng(6)
    mixes = (None, ("runs", "shortruns", "tiny"), ("runs", "sparse", "dense"), ("full", "nearfull", "runs", "verydense"))
    for it in range(iters):
        n = int(rng.integers(0, 9))
        profs = mixes[it % len(mixes)]
        kw = dict(max_keys=6, key_space=8) if profs is None else dict(max_keys=4, key_space=4, profiles=profs)
        hs = [ref.from_sorted(random_bitmap(rng, **kw)) for _ in range(n)]
        arr = (C.c_void_p * max(n, 1))(*hs)
        want_or, want_heap, want_xor = ref.or_many(hs), ref.or_many_heap(hs), ref.xor_many(hs)
        for nm, fn, want_bytes, want_set in (("or_many", hip.roaring_bitmap_or_many, want_or, want_or),
                                             ("or_many_heap", hip.roaring_bitmap_or_many_heap, want_heap, want_heap),
                                             ("xor_many", hip.roaring_bitmap_xor_many, want_xor, want_xor)):
            got = fn(n, arr)
            assert got, "drop-in many-way aggregation returned NULL (no device?)"
            assert ref.validate(got), (it, nm)
            assert np.array_equal(ref.to_array(got), ref.to_array(want_set)), (it, nm)
            assert ref.serialize(got) == ref.serialize(want_bytes), (it, nm, "bytes")
            ref.free(got)
        for h in hs + [want_or, want_heap, want_xor]:
            ref.free(h)


def test_dropin_many(hip, ref):
    dropin_many_body(hip, ref)


def alloc_failure_body(hip, ref, child_cmd):
    """The void in-place drop-ins under a device allocation that fails (rhip_debug_fail_allocs): a TRANSIENT failure is
    survived -- the drop-ins release what they hold and the call is made once more, the result is the reference's --
    and only a failure that persists ends the process, loudly (the reference's in-place functions have no error channel,
    roaring.h:280-348; leaving x1 unchanged would be a silently wrong result)."""
    import subprocess, sys
    hip.rhip_debug_fail_allocs.restype = None
    hip.rhip_debug_fail_allocs.argtypes = [C.c_int, C.c_int]
    rng = np.random.default_rng(12)

    def grown(k):  # operands larger than anything the lanes have recycled so far: the call HAS to allocate
        n = 60 * (k + 1)
        keys = np.sort(rng.choice(4000, n, replace=False)).astype(np.uint32)
        return np.concatenate([(kk << np.uint32(16)) | np.unique(rng.integers(0, 65536, 300 + 50 * k)).astype(np.uint32) for kk in keys])
    for it in range(6):
        a, b = ref.from_sorted(grown(it)), ref.from_sorted(grown(it))
        want = ref.op("or", a, b)
        hip.rhip_debug_fail_allocs(it % 3, 1)  # the first / second / third allocation of the call fails, once
        hip.roaring_bitmap_or_inplace(a, b)
        hip.rhip_debug_fail_allocs(0, 0)
        assert ref.validate(a) and ref.serialize(a) == ref.serialize(want), it
        for h in (a, b, want):
            ref.free(h)
    # a pointer-returning drop-in reports the same failure as NULL (roaring.h:216-226) and the next call works again
    a, b = ref.from_sorted(grown(8)), ref.from_sorted(grown(8))
    hip.rhip_debug_fail_allocs(0, 1000)
    assert not hip.roaring_bitmap_or(a, b)
    hip.rhip_debug_fail_allocs(0, 0)
    r = hip.roaring_bitmap_or(a, b)
    w = ref.op("or", a, b)
    assert r and ref.serialize(r) == ref.serialize(w)
    for h in (a, b, r, w):
        ref.free(h)
    # persistent failure inside a void function: abort, with the reason on stderr (a process of its own)
    p = subprocess.run(child_cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "has no way to report it" in p.stderr, (p.returncode, p.stderr[-2000:])


def test_inplace_dropins_survive_transient_alloc_failure(hip, ref):
    import os, sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hook_child.py")
    alloc_failure_body(hip, ref, [sys.executable, child, "hip", "allocfail"])


def test_dropin_lazy_family(hip, ref):
    """lazy_or / lazy_xor are eager on the device; repair_after_lazy must also canonicalise a bitmap that
    the REFERENCE's lazy functions left unrepaired (unknown bitset cardinalities, non-efficient runs)."""
    vp = C.c_void_p
    hip.roaring_bitmap_lazy_or.restype = vp; hip.roaring_bitmap_lazy_or.argtypes = [vp, vp, C.c_bool]
    hip.roaring_bitmap_lazy_xor.restype = vp; hip.roaring_bitmap_lazy_xor.argtypes = [vp, vp]
    hip.roaring_bitmap_lazy_or_inplace.restype = None; hip.roaring_bitmap_lazy_or_inplace.argtypes = [vp, vp, C.c_bool]
    hip.roaring_bitmap_lazy_xor_inplace.restype = None; hip.roaring_bitmap_lazy_xor_inplace.argtypes = [vp, vp]
    hip.roaring_bitmap_repair_after_lazy.restype = None; hip.roaring_bitmap_repair_after_lazy.argtypes = [vp]
    R = ref.L
    R.roaring_bitmap_lazy_or.restype = vp; R.roaring_bitmap_lazy_or.argtypes = [vp, vp, C.c_bool]
    R.roaring_bitmap_lazy_xor.restype = vp; R.roaring_bitmap_lazy_xor.argtypes = [vp, vp]
    R.roaring_bitmap_lazy_or_inplace.restype = None; R.roaring_bitmap_lazy_or_inplace.argtypes = [vp, vp, C.c_bool]
    rng = np.random.default_rng(8)
    for it in range(15):
        a, b, c = (ref.from_sorted(random_bitmap(rng)) for _ in range(3))
        want_or, want_xor = ref.op("or", a, b), ref.op("xor", a, b)
        got = hip.roaring_bitmap_lazy_or(a, b, bool(it & 1))
        assert got, "drop-in roaring_bitmap_lazy_or returned NULL (no device?)"
        assert ref.validate(got) and ref.serialize(got) == ref.serialize(want_or)
        gx = hip.roaring_bitmap_lazy_xor(a, b)
        assert gx, "drop-in roaring_bitmap_lazy_xor returned NULL (no device?)"
        assert ref.validate(gx) and ref.serialize(gx) == ref.serialize(want_xor)
        # reference-made lazy bitmap (a | b | c, unrepaired) repaired by OUR repair_after_lazy
        lazy = R.roaring_bitmap_lazy_or(a, b, bool(it & 1))
        R.roaring_bitmap_lazy_or_inplace(lazy, c, bool(it & 1))
        hip.roaring_bitmap_repair_after_lazy(lazy)
        abc = ref.op("or", want_or, c)
        assert ref.validate(lazy)
        assert np.array_equal(ref.to_array(lazy), ref.to_array(abc)), it
        lx = R.roaring_bitmap_lazy_xor(a, b)
        hip.roaring_bitmap_repair_after_lazy(lx)
        assert ref.validate(lx) and np.array_equal(ref.to_array(lx), ref.to_array(want_xor))
        for h in (a, b, c, want_or, want_xor, got, gx, lazy, abc, lx):
            ref.free(h)
