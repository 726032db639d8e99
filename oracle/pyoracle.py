"""TEST INFRASTRUCTURE ONLY -- ctypes front-ends for the two CPU checkers.

* ``Oracle`` : oracle/liboracle.so, the plain-C restatement (roaring_oracle.c).
* ``Ref``    : oracle/_ref/libcroaring_ref.so, the REAL CRoaring 5.1.0 library
               compiled from /root/reference by oracle/Makefile (present only
               where it was prebuilt; ``Ref.available()`` says so).

Both expose the same small interface over opaque handles so tests can run the
same assertions against either.  Nothing under ``croaring_amd/`` imports this
module; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline do.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OPS = {"and": 0, "or": 1, "xor": 2, "andnot": 3}


def build(quiet: bool = True) -> None:
    """Compile liboracle.so (and _ref when /root/reference is present)."""
    subprocess.run(["make", "-C", HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


class Oracle:
    """The C restatement.  Handles are ``oc_bitmap_t*`` (c_void_p)."""

    name = "oracle"

    def __init__(self):
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
        L.oc_deserialize.restype = vp; L.oc_deserialize.argtypes = [C.c_char_p, sz]
        L.oc_size_in_bytes.restype = sz; L.oc_size_in_bytes.argtypes = [vp]
        L.oc_serialize.restype = sz; L.oc_serialize.argtypes = [vp, C.c_char_p]
        L.oc_frozen_size_in_bytes.restype = sz; L.oc_frozen_size_in_bytes.argtypes = [vp]
        L.oc_frozen_serialize.restype = sz; L.oc_frozen_serialize.argtypes = [vp, C.c_char_p]
        L.oc_frozen_deserialize.restype = vp; L.oc_frozen_deserialize.argtypes = [C.c_char_p, sz]
        L.oc_op.restype = vp; L.oc_op.argtypes = [C.c_int, vp, vp]
        L.oc_get_cardinality.restype = u64; L.oc_get_cardinality.argtypes = [vp]
        L.oc_and_cardinality.restype = u64; L.oc_and_cardinality.argtypes = [vp, vp]
        L.oc_op_cardinality.restype = u64; L.oc_op_cardinality.argtypes = [C.c_int, vp, vp]
        L.oc_or_many.restype = vp; L.oc_or_many.argtypes = [sz, C.POINTER(vp)]
        L.oc_or_many_heap.restype = vp; L.oc_or_many_heap.argtypes = [sz, C.POINTER(vp)]
        L.oc_xor_many.restype = vp; L.oc_xor_many.argtypes = [sz, C.POINTER(vp)]
        L.oc_free.restype = None; L.oc_free.argtypes = [vp]
        L.oc_from_sorted.restype = vp; L.oc_from_sorted.argtypes = [vp, sz]
        L.oc_run_optimize.restype = C.c_int; L.oc_run_optimize.argtypes = [vp]
        L.oc_remove_run_compression.restype = C.c_int; L.oc_remove_run_compression.argtypes = [vp]
        L.oc_flip.restype = vp; L.oc_flip.argtypes = [vp, u64, u64]
        for nm in ("oc_intersect", "oc_is_subset", "oc_is_strict_subset", "oc_equals"):
            f = getattr(L, nm); f.restype = C.c_int; f.argtypes = [vp, vp]
        L.oc_validate.restype = C.c_int; L.oc_validate.argtypes = [vp]
        L.oc_to_uint32.restype = None; L.oc_to_uint32.argtypes = [vp, vp]
        L.oc_type_counts.restype = None; L.oc_type_counts.argtypes = [vp, vp]
        L.oc64_deserialize.restype = vp; L.oc64_deserialize.argtypes = [C.c_char_p, sz]
        L.oc64_size_in_bytes.restype = sz; L.oc64_size_in_bytes.argtypes = [vp]
        L.oc64_serialize.restype = sz; L.oc64_serialize.argtypes = [vp, C.c_char_p]
        L.oc64_from_sorted.restype = vp; L.oc64_from_sorted.argtypes = [vp, sz]
        L.oc64_run_optimize.restype = C.c_int; L.oc64_run_optimize.argtypes = [vp]
        L.oc64_op.restype = vp; L.oc64_op.argtypes = [C.c_int, vp, vp]
        L.oc64_get_cardinality.restype = u64; L.oc64_get_cardinality.argtypes = [vp]
        L.oc64_or_many.restype = vp; L.oc64_or_many.argtypes = [sz, C.POINTER(vp)]
        L.oc64_free.restype = None; L.oc64_free.argtypes = [vp]
        L.oc64_flip.restype = vp; L.oc64_flip.argtypes = [vp, u64, u64]
        self.L = L

    # -- 32-bit
    def deserialize(self, buf: bytes):
        h = self.L.oc_deserialize(buf, len(buf))
        if not h:
            raise ValueError("oracle: bad portable buffer")
        return h

    def serialize(self, h) -> bytes:
        n = self.L.oc_size_in_bytes(h)
        out = C.create_string_buffer(n)
        w = self.L.oc_serialize(h, out)
        assert w == n
        return out.raw

    def frozen_serialize(self, h) -> bytes:
        n = self.L.oc_frozen_size_in_bytes(h)
        out = C.create_string_buffer(n)
        assert self.L.oc_frozen_serialize(h, out) == n
        return out.raw

    def frozen_deserialize(self, buf: bytes):
        """roaring_bitmap_frozen_view's acceptance, as a copy (None when the view would be NULL)."""
        return self.L.oc_frozen_deserialize(buf, len(buf)) or None

    def from_sorted(self, vals, run_optimize=True):
        v = np.ascontiguousarray(vals, dtype=np.uint32)
        h = self.L.oc_from_sorted(v.ctypes.data, v.size)
        if run_optimize:
            self.L.oc_run_optimize(h)
        return h

    def run_optimize(self, h) -> bool:
        return bool(self.L.oc_run_optimize(h))

    def remove_run_compression(self, h) -> bool:
        return bool(self.L.oc_remove_run_compression(h))

    def flip(self, h, start: int, end: int):
        return self.L.oc_flip(h, start, end)

    def predicate(self, pred, a, b) -> bool:
        return bool(getattr(self.L, {"intersect": "oc_intersect", "is_subset": "oc_is_subset",
                                     "is_strict_subset": "oc_is_strict_subset", "equals": "oc_equals"}[pred])(a, b))

    def op(self, op, a, b):
        return self.L.oc_op(OPS[op], a, b)

    def cardinality(self, h) -> int:
        return self.L.oc_get_cardinality(h)

    def op_cardinality(self, op, a, b) -> int:
        return self.L.oc_op_cardinality(OPS[op], a, b)

    def _many(self, fn, hs):
        arr = (C.c_void_p * len(hs))(*hs)
        return fn(len(hs), arr)

    def or_many(self, hs):
        return self._many(self.L.oc_or_many, hs)

    def xor_many(self, hs):
        return self._many(self.L.oc_xor_many, hs)

    def or_many_heap(self, hs):
        return self._many(self.L.oc_or_many_heap, hs)

    def validate(self, h) -> bool:
        return bool(self.L.oc_validate(h))

    def to_array(self, h) -> np.ndarray:
        out = np.empty(self.cardinality(h), dtype=np.uint32)
        self.L.oc_to_uint32(h, out.ctypes.data)
        return out

    def type_counts(self, h):
        out = np.zeros(3, dtype=np.int64)
        self.L.oc_type_counts(h, out.ctypes.data)
        return tuple(int(x) for x in out)  # (bitset, array, run)

    def free(self, h):
        self.L.oc_free(h)

    # -- 64-bit
    def deserialize64(self, buf: bytes):
        h = self.L.oc64_deserialize(buf, len(buf))
        if not h:
            raise ValueError("oracle: bad portable64 buffer")
        return h

    def serialize64(self, h) -> bytes:
        n = self.L.oc64_size_in_bytes(h)
        out = C.create_string_buffer(n)
        assert self.L.oc64_serialize(h, out) == n
        return out.raw

    def from_sorted64(self, vals, run_optimize=True):
        v = np.ascontiguousarray(vals, dtype=np.uint64)
        h = self.L.oc64_from_sorted(v.ctypes.data, v.size)
        if run_optimize:
            self.L.oc64_run_optimize(h)
        return h

    def op64(self, op, a, b):
        return self.L.oc64_op(OPS[op], a, b)

    def cardinality64(self, h) -> int:
        return self.L.oc64_get_cardinality(h)

    def or_many64(self, hs):
        return self._many(self.L.oc64_or_many, hs)

    def flip64(self, h, start: int, end: int):
        return self.L.oc64_flip(h, start, end)

    def free64(self, h):
        self.L.oc64_free(h)


class Ref:
    """The real CRoaring library (oracle/_ref).  Handles are ``roaring_bitmap_t*``."""

    name = "reference"
    PATH = os.path.join(HERE, "_ref", "libcroaring_ref.so")

    @classmethod
    def available(cls) -> bool:
        return os.path.exists(cls.PATH)

    def __init__(self):
        L = C.CDLL(self.PATH)
        vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
        for nm in ("and", "or", "xor", "andnot"):
            f = getattr(L, f"roaring_bitmap_{nm}"); f.restype = vp; f.argtypes = [vp, vp]
            f = getattr(L, f"roaring_bitmap_{nm}_cardinality"); f.restype = u64; f.argtypes = [vp, vp]
            f = getattr(L, f"roaring64_bitmap_{nm}"); f.restype = vp; f.argtypes = [vp, vp]
        L.roaring_bitmap_portable_deserialize_safe.restype = vp
        L.roaring_bitmap_portable_deserialize_safe.argtypes = [C.c_char_p, sz]
        L.roaring_bitmap_portable_size_in_bytes.restype = sz
        L.roaring_bitmap_portable_size_in_bytes.argtypes = [vp]
        L.roaring_bitmap_portable_serialize.restype = sz
        L.roaring_bitmap_portable_serialize.argtypes = [vp, C.c_char_p]
        L.roaring_bitmap_frozen_size_in_bytes.restype = sz; L.roaring_bitmap_frozen_size_in_bytes.argtypes = [vp]
        L.roaring_bitmap_frozen_serialize.restype = None; L.roaring_bitmap_frozen_serialize.argtypes = [vp, vp]
        L.roaring_bitmap_frozen_view.restype = vp; L.roaring_bitmap_frozen_view.argtypes = [vp, sz]
        L.roaring_bitmap_copy.restype = vp; L.roaring_bitmap_copy.argtypes = [vp]
        L.roaring_bitmap_get_cardinality.restype = u64; L.roaring_bitmap_get_cardinality.argtypes = [vp]
        L.roaring_bitmap_or_many.restype = vp; L.roaring_bitmap_or_many.argtypes = [sz, C.POINTER(vp)]
        L.roaring_bitmap_xor_many.restype = vp; L.roaring_bitmap_xor_many.argtypes = [sz, C.POINTER(vp)]
        L.roaring_bitmap_or_many_heap.restype = vp
        L.roaring_bitmap_or_many_heap.argtypes = [C.c_uint32, C.POINTER(vp)]
        L.roaring_bitmap_free.restype = None; L.roaring_bitmap_free.argtypes = [vp]
        L.roaring_bitmap_of_ptr.restype = vp; L.roaring_bitmap_of_ptr.argtypes = [sz, vp]
        L.roaring_bitmap_run_optimize.restype = C.c_bool; L.roaring_bitmap_run_optimize.argtypes = [vp]
        L.roaring_bitmap_shrink_to_fit.restype = sz; L.roaring_bitmap_shrink_to_fit.argtypes = [vp]
        L.roaring_bitmap_remove_run_compression.restype = C.c_bool
        L.roaring_bitmap_remove_run_compression.argtypes = [vp]
        L.roaring_bitmap_flip.restype = vp; L.roaring_bitmap_flip.argtypes = [vp, u64, u64]
        for nm in ("intersect", "is_subset", "is_strict_subset", "equals"):
            f = getattr(L, f"roaring_bitmap_{nm}"); f.restype = C.c_bool; f.argtypes = [vp, vp]
        L.roaring_bitmap_to_uint32_array.restype = None; L.roaring_bitmap_to_uint32_array.argtypes = [vp, vp]
        L.roaring_bitmap_internal_validate.restype = C.c_bool
        L.roaring_bitmap_internal_validate.argtypes = [vp, C.POINTER(C.c_char_p)]
        L.roaring64_bitmap_portable_deserialize_safe.restype = vp
        L.roaring64_bitmap_portable_deserialize_safe.argtypes = [C.c_char_p, sz]
        L.roaring64_bitmap_portable_size_in_bytes.restype = sz
        L.roaring64_bitmap_portable_size_in_bytes.argtypes = [vp]
        L.roaring64_bitmap_portable_serialize.restype = sz
        L.roaring64_bitmap_portable_serialize.argtypes = [vp, C.c_char_p]
        L.roaring64_bitmap_get_cardinality.restype = u64; L.roaring64_bitmap_get_cardinality.argtypes = [vp]
        L.roaring64_bitmap_of_ptr.restype = vp; L.roaring64_bitmap_of_ptr.argtypes = [sz, vp]
        L.roaring64_bitmap_run_optimize.restype = C.c_bool; L.roaring64_bitmap_run_optimize.argtypes = [vp]
        L.roaring64_bitmap_free.restype = None; L.roaring64_bitmap_free.argtypes = [vp]
        L.roaring64_bitmap_or_inplace.restype = None; L.roaring64_bitmap_or_inplace.argtypes = [vp, vp]
        L.roaring64_bitmap_create.restype = vp; L.roaring64_bitmap_create.argtypes = []
        L.roaring64_bitmap_flip.restype = vp; L.roaring64_bitmap_flip.argtypes = [vp, u64, u64]
        self.L = L

    def deserialize(self, buf: bytes):
        h = self.L.roaring_bitmap_portable_deserialize_safe(buf, len(buf))
        if not h:
            raise ValueError("reference: bad portable buffer")
        return h

    def serialize(self, h) -> bytes:
        n = self.L.roaring_bitmap_portable_size_in_bytes(h)
        out = C.create_string_buffer(n)
        assert self.L.roaring_bitmap_portable_serialize(h, out) == n
        return out.raw

    def frozen_serialize(self, h) -> bytes:
        n = self.L.roaring_bitmap_frozen_size_in_bytes(h)
        out = C.create_string_buffer(n)
        self.L.roaring_bitmap_frozen_serialize(h, C.addressof(out))
        return out.raw

    def frozen_deserialize(self, buf: bytes):
        """roaring_bitmap_frozen_view on a 32-byte aligned copy of buf, then roaring_bitmap_copy (the view borrows the
        buffer); None when the view is NULL."""
        raw = np.zeros(len(buf) + 64, dtype=np.uint8)
        o = (-raw.ctypes.data) % 32
        raw[o:o + len(buf)] = np.frombuffer(buf, dtype=np.uint8)
        v = self.L.roaring_bitmap_frozen_view(raw.ctypes.data + o, len(buf))
        if not v:
            return None
        h = self.L.roaring_bitmap_copy(v)
        self.L.roaring_bitmap_free(v)
        return h

    def from_sorted(self, vals, run_optimize=True):
        # benchmarks/benchmark.cpp:1938-1942: of_ptr + run_optimize + shrink_to_fit
        v = np.ascontiguousarray(vals, dtype=np.uint32)
        h = self.L.roaring_bitmap_of_ptr(v.size, v.ctypes.data)
        if run_optimize:
            self.L.roaring_bitmap_run_optimize(h)
        self.L.roaring_bitmap_shrink_to_fit(h)
        return h

    def run_optimize(self, h) -> bool:
        return bool(self.L.roaring_bitmap_run_optimize(h))

    def remove_run_compression(self, h) -> bool:
        return bool(self.L.roaring_bitmap_remove_run_compression(h))

    def flip(self, h, start: int, end: int):
        return self.L.roaring_bitmap_flip(h, start, end)

    def predicate(self, pred, a, b) -> bool:
        return bool(getattr(self.L, f"roaring_bitmap_{pred}")(a, b))

    def op(self, op, a, b):
        return getattr(self.L, f"roaring_bitmap_{op}")(a, b)

    def cardinality(self, h) -> int:
        return self.L.roaring_bitmap_get_cardinality(h)

    def op_cardinality(self, op, a, b) -> int:
        return getattr(self.L, f"roaring_bitmap_{op}_cardinality")(a, b)

    def _arr(self, hs):
        return (C.c_void_p * len(hs))(*hs)

    def or_many(self, hs):
        return self.L.roaring_bitmap_or_many(len(hs), self._arr(hs))

    def xor_many(self, hs):
        return self.L.roaring_bitmap_xor_many(len(hs), self._arr(hs))

    def or_many_heap(self, hs):
        return self.L.roaring_bitmap_or_many_heap(len(hs), self._arr(hs))

    def validate(self, h) -> bool:
        reason = C.c_char_p()
        return bool(self.L.roaring_bitmap_internal_validate(h, C.byref(reason)))

    def to_array(self, h) -> np.ndarray:
        out = np.empty(self.cardinality(h), dtype=np.uint32)
        self.L.roaring_bitmap_to_uint32_array(h, out.ctypes.data)
        return out

    def free(self, h):
        self.L.roaring_bitmap_free(h)

    # -- 64-bit
    def deserialize64(self, buf: bytes):
        h = self.L.roaring64_bitmap_portable_deserialize_safe(buf, len(buf))
        if not h:
            raise ValueError("reference: bad portable64 buffer")
        return h

    def serialize64(self, h) -> bytes:
        n = self.L.roaring64_bitmap_portable_size_in_bytes(h)
        out = C.create_string_buffer(n)
        assert self.L.roaring64_bitmap_portable_serialize(h, out) == n
        return out.raw

    def from_sorted64(self, vals, run_optimize=True):
        v = np.ascontiguousarray(vals, dtype=np.uint64)
        h = self.L.roaring64_bitmap_of_ptr(v.size, v.ctypes.data)
        if run_optimize:
            self.L.roaring64_bitmap_run_optimize(h)
        return h

    def op64(self, op, a, b):
        return getattr(self.L, f"roaring64_bitmap_{op}")(a, b)

    def cardinality64(self, h) -> int:
        return self.L.roaring64_bitmap_get_cardinality(h)

    def flip64(self, h, start: int, end: int):
        return self.L.roaring64_bitmap_flip(h, start, end)

    def or_many64(self, hs):
        # no C many-way API for 64-bit: left fold of or_inplace
        acc = self.L.roaring64_bitmap_create()
        for h in hs:
            self.L.roaring64_bitmap_or_inplace(acc, h)
        return acc

    def free64(self, h):
        self.L.roaring64_bitmap_free(h)


def best_checker():
    """The real reference when its prebuilt .so is present, else the restatement."""
    return Ref() if Ref.available() else Oracle()
