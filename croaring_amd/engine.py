"""Host-side mirror of the CRoaring hot-path API over device-resident bitmaps.

Vocabulary follows the reference (include/roaring/roaring.h): bitmaps, containers, portable
serialization, and / or / xor / andnot, *_cardinality, or_many / xor_many.  A `Pool` is an
immutable set of bitmaps living in HBM; operations take index arrays into pools and return new
pools, so chained expressions never leave the device.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import ClassStats, RoaringHipError, Stats, Partials

OPS = {"and": 0, "or": 1, "xor": 2, "andnot": 3}
PREDS = {"intersect": 0, "is_subset": 1, "is_strict_subset": 2, "equals": 3}


def _u32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def _pair_ids(lhs, rhs):
    """Validated (lhs, rhs) index arrays of a pairwise batch: both given, one-dimensional, equally long."""
    if lhs is None or rhs is None:
        raise ValueError("pairwise batches need both lhs and rhs index arrays")
    lhs, rhs = _u32(lhs).ravel(), _u32(rhs).ravel()
    if lhs.shape != rhs.shape:
        raise ValueError("lhs/rhs length mismatch")
    return lhs, rhs


def synth_sparse_portable(first: int, stride: int, count: int):
    """SURVEY §8d C4 generator (rhip_synth_sparse_sizes / _fill): portable images of the sparse bitmaps first,
    first + stride, ... packed back to back -> (uint8 blob, uint64 offsets[count + 1]).  Host only, no device."""
    lib = _lib.load()
    offs = np.zeros(count + 1, dtype=np.uint64)
    if lib.rhip_synth_sparse_sizes(first, stride, count, offs.ctypes.data) != 0:
        raise RoaringHipError("synth_sparse_sizes failed: " + _lib.last_error())
    blob = np.empty(int(offs[count]), dtype=np.uint8)
    if count and lib.rhip_synth_sparse_fill(first, stride, count, offs.ctypes.data, blob.ctypes.data) != 0:
        raise RoaringHipError("synth_sparse_fill failed: " + _lib.last_error())
    return blob, offs


class Engine:
    """One per process and device (rhip_ctx_t): owns the HIP stream and scratch buffers."""

    def __init__(self, device: int = -1):
        self.lib = _lib.load()
        self.h = self.lib.rhip_ctx_create(device)
        if not self.h:
            raise RoaringHipError("rhip_ctx_create failed: " + self._err())

    def _err(self) -> str:
        return (self.lib.rhip_last_error() or b"").decode()

    def close(self):
        if getattr(self, "h", None):
            self.lib.rhip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self) -> int:
        return self.lib.rhip_ctx_stream(self.h)

    def synchronize(self):
        if self.lib.rhip_ctx_synchronize(self.h) != 0:
            raise RoaringHipError(self._err())

    def host_clock(self, reset: bool = True) -> list:
        """Host microseconds of rhip_pairwise by phase since the last reset (include/roaring_hip.h)."""
        out = (C.c_double * 8)()
        self.lib.rhip_debug_host_clock(self.h, out, 1 if reset else 0)
        return list(out)

    def trim(self) -> int:
        """Release the spare result arenas kept from placement searches (rhip_ctx_trim); bytes released."""
        return int(self.lib.rhip_ctx_trim(self.h))

    def join_recovered(self) -> int:
        """Batches of this context whose flag join gave up and that were finished through the event fallback."""
        return int(self.lib.rhip_debug_join_recovered(self.h))

    def plan_cached(self) -> bool:
        """Did the last batch begun on this engine take its plan from its pair list's cache?"""
        return bool(self.lib.rhip_debug_plan_cached(self.h))

    def last_placement(self) -> list:
        """Probe rates (GB/s) of the last measured placement of a result arena (rhip_debug_last_placement): single chunks and
        compositions, candidate allocations, address positions, in the order they were probed."""
        out = (C.c_float * 128)()
        n = self.lib.rhip_debug_last_placement(self.h, out, 128)
        return [round(float(out[k]), 1 if out[k] >= 10 else 6) for k in range(min(n, 128))]  # (the emulator's rates are ~0.05 GB/s)

    def set_timing(self, on: bool):
        self.lib.rhip_ctx_set_timing(self.h, 1 if on else 0)

    def set_class_stats(self, on: bool):
        """Per-kernel algorithmic bytes of every pairwise batch (one more kernel + wait per batch while on)."""
        self.lib.rhip_ctx_set_class_stats(self.h, 1 if on else 0)

    def last_class_stats(self) -> dict:
        """{kernel: {"items", "bytes_in", "bytes_out"}} of the last pairwise batch ended (set_class_stats(True))."""
        buf = (ClassStats * 16)()
        n = self.lib.rhip_last_class_stats(self.h, buf, 16)
        return {buf[k].kernel.decode(): {"items": int(buf[k].items), "bytes_in": int(buf[k].bytes_in),
                                         "bytes_out": int(buf[k].bytes_out)} for k in range(min(n, 16))}

    def last_stats(self) -> dict:
        s = Stats()
        self.lib.rhip_last_stats(self.h, C.byref(s))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    # ---- loading -------------------------------------------------------
    def _from_bufs(self, fn, bufs: Sequence[bytes]) -> "Pool":
        n = len(bufs)
        arr = (C.c_char_p * max(n, 1))(*bufs)
        lens = (C.c_size_t * max(n, 1))(*[len(b) for b in bufs])
        h = fn(self.h, n, arr, lens)
        if not h:
            raise RoaringHipError("deserialize failed: " + self._err())
        return Pool(self, h)

    def pool_from_serialized(self, bufs: Sequence[bytes]) -> "Pool":
        """roaring_bitmap_portable_deserialize_safe for every buffer, into one HBM pool."""
        return self._from_bufs(self.lib.rhip_pool_from_portable, bufs)

    def pool_from_packed(self, blob: np.ndarray, offsets, lens, is64: bool = False) -> "Pool":
        """Same as pool_from_serialized, for n portable bitmaps packed back to back in one uint8 array
        (no per-bitmap Python objects: used for 10^5-bitmap pools)."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint64)
        ptrs = (offsets + np.uint64(blob.ctypes.data)).astype(np.uint64)
        fn = self.lib.rhip_pool_from_portable64 if is64 else self.lib.rhip_pool_from_portable
        h = fn(self.h, len(offsets), ptrs.ctypes.data, lens.ctypes.data)
        if not h:
            raise RoaringHipError("deserialize failed: " + self._err())
        return Pool(self, h)

    def pool_from_blob(self, blob, offsets, lens=None, is64: bool = False) -> "Pool":
        """n portable images packed in one uint8 array (image i = blob[offsets[i] : offsets[i] + lens[i]]; lens
        defaults to the gaps of an (n+1)-entry offsets array, i.e. the output of Pool.serialize_many): ONE upload,
        parsed and validated on the device."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        if lens is None:
            lens = np.diff(offsets)
            offsets = np.ascontiguousarray(offsets[:-1])
        lens = np.ascontiguousarray(lens, dtype=np.uint64)
        if offsets.shape != lens.shape:
            raise ValueError("offsets/lens length mismatch")
        h = self.lib.rhip_pool_from_blob(self.h, blob.ctypes.data, blob.size, offsets.size, offsets.ctypes.data,
                                         lens.ctypes.data, 1 if is64 else 0)
        if not h:
            raise RoaringHipError("deserialize failed: " + self._err())
        return Pool(self, h)

    def pool_from_frozen(self, blob, offsets, lens) -> "Pool":
        """n FROZEN images (roaring_bitmap_frozen_serialize) in one uint8 array, image i = blob[offsets[i] : offsets[i] +
        lens[i]] -- e.g. the output of Pool.frozen_serialize_many: ONE upload, parsed and validated on the device."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.uint64)
        offsets = np.ascontiguousarray(np.asarray(offsets, dtype=np.uint64)[:lens.size])
        if offsets.shape != lens.shape:
            raise ValueError("offsets/lens length mismatch")
        h = self.lib.rhip_pool_from_frozen(self.h, blob.ctypes.data, blob.size, offsets.size, offsets.ctypes.data,
                                           lens.ctypes.data)
        if not h:
            raise RoaringHipError("frozen load failed: " + self._err())
        return Pool(self, h)

    def pool_from_values(self, lists: Sequence, is64: bool = False) -> "Pool":
        """roaring_bitmap_of_ptr for every list, on the device; each list sorted and free of duplicates."""
        dt = np.uint64 if is64 else np.uint32
        arrs = [np.ascontiguousarray(v, dtype=dt) for v in lists]
        offs = np.zeros(len(arrs) + 1, dtype=np.uint64)
        if arrs:
            offs[1:] = np.cumsum([a.size for a in arrs])
        vals = np.concatenate(arrs) if arrs else np.zeros(0, dt)
        return self.pool_from_packed_values(vals, offs, is64)

    def pool_from_packed_values(self, values, offsets, is64: bool = False) -> "Pool":
        values = np.ascontiguousarray(values, dtype=np.uint64 if is64 else np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        fn = self.lib.rhip_pool_from_sorted_u64 if is64 else self.lib.rhip_pool_from_sorted_u32
        h = fn(self.h, offsets.size - 1, values.ctypes.data, offsets.ctypes.data)
        if not h:
            raise RoaringHipError("from_values failed: " + self._err())
        return Pool(self, h)

    def pool_from_serialized64(self, bufs: Sequence[bytes]) -> "Pool":
        """roaring64_bitmap_portable_deserialize_safe for every buffer."""
        return self._from_bufs(self.lib.rhip_pool_from_portable64, bufs)

    def pool_synth_bitset(self, n_bitmaps: int, n_containers: int, seed: int) -> "Pool":
        h = self.lib.rhip_pool_synth_bitset(self.h, n_bitmaps, n_containers, seed & (2**64 - 1))
        if not h:
            raise RoaringHipError("synth failed: " + self._err())
        return Pool(self, h)

    # ---- pairwise ------------------------------------------------------
    def pairwise(self, op: str, A: "Pool", lhs, B: Optional["Pool"] = None, rhs=None,
                 reuse: Optional["Pool"] = None) -> "Pool":
        """result[k] = op(A[lhs[k]], B[rhs[k]]) -- roaring_bitmap_{and,or,xor,andnot} batched."""
        B = A if B is None else B
        lhs, rhs = _pair_ids(lhs, rhs)
        rh = None
        if reuse is not None:
            rh, reuse.h = reuse.h, None  # consumed
        h = self.lib.rhip_pairwise(self.h, OPS[op], A.h, B.h, lhs.size, lhs.ctypes.data, rhs.ctypes.data, rh)
        if not h:
            err = self._err()
            if reuse is not None and "still in flight" in err:
                reuse.h = rh  # refused, not consumed (include/roaring_hip.h): the caller's Pool stays valid
            raise RoaringHipError(f"pairwise {op} failed: " + err)
        return Pool(self, h)

    def pairwise_placed(self, op: str, A: "Pool", lhs, B: Optional["Pool"] = None, rhs=None, tries: int = 12,
                        keep: int = 1, timing_after: bool = False):
        """Warm-up helper for a service that recycles large result pools (`reuse=`): run the batch `tries` times, each
        into a freshly allocated result pool while the earlier ones are still alive, and keep the `keep` pools whose
        kernels ran fastest (a result pool serves any op).  Returns (list of `keep` result pools, fastest first,
        [device ms of every try]).

        Why: the time of a streaming kernel over multi-GiB operands depends on which PHYSICAL pages the driver hands
        the result arena -- the same virtual address re-allocated gives the bitset kernel of BASELINE config C2
        4.37-4.45 ms or 4.64-4.71 ms per 250-pair batch, at random and for the life of the allocation, whatever the
        arena's offset, alignment or size rounding (scripts/arena_skew_sweep.py).  Nothing in the address
        predicts the mode, and consecutive allocations come in streaks of one mode (eight in a row have been seen), so
        the only handle is to look: a dozen allocations at start-up, the best kept.
        (Uses the HIP-event timing of the context; leaves it `timing_after`.)"""
        self.set_timing(True)
        cands = []
        for _ in range(max(keep, tries)):
            try:
                res = self.pairwise(op, A, lhs, B, rhs)  # (first touch of a fresh arena)
            except RoaringHipError:
                if len(cands) >= keep:  # (out of device memory for another candidate: choose among those there are)
                    break
                raise
            ms = []
            for _ in range(2):
                res = self.pairwise(op, A, lhs, B, rhs, reuse=res)
                st = self.last_stats()
                ms.append(st["ms_bitset_kernel"] if st["ms_bitset_kernel"] > 0 else st["ms_total"])
            cands.append((min(ms), res))
        order = sorted(range(len(cands)), key=lambda k: cands[k][0])
        for k in order[keep:]:
            cands[k][1].free()
        self.set_timing(timing_after)
        return [cands[k][1] for k in order[:keep]], [round(c[0], 4) for c in cands]

    def pairwise_multi(self, ops: Sequence[str], A: "Pool", lhs, B: Optional["Pool"] = None, rhs=None,
                       reuse: Optional["Pool"] = None) -> "Pool":
        """Several ops over ONE pair list in one batch (rhip_pairwise_multi): result bitmap o * len(lhs) + k =
        ops[o](A[lhs[k]], B[rhs[k]]) -- the and / or / xor / andnot sweep of the reference benchmark planned once."""
        B = A if B is None else B
        lhs, rhs = _pair_ids(lhs, rhs)
        codes = (C.c_int * len(ops))(*[OPS[o] for o in ops])
        rh = None
        if reuse is not None:
            rh, reuse.h = reuse.h, None  # consumed
        h = self.lib.rhip_pairwise_multi(self.h, len(ops), codes, A.h, B.h, lhs.size, lhs.ctypes.data, rhs.ctypes.data, rh)
        if not h:
            err = self._err()
            if reuse is not None and "still in flight" in err:
                reuse.h = rh
            raise RoaringHipError(f"pairwise_multi {list(ops)} failed: " + err)
        return Pool(self, h)

    def pairwise_begin(self, op: str, A: "Pool", lhs, B: Optional["Pool"] = None, rhs=None,
                       reuse: Optional["Pool"] = None) -> "Batch":
        """First half of `pairwise`: enqueue the batch and return without waiting for the device (rhip_pairwise_begin).
        `Batch.end()` returns the result pool.  Up to 4 batches may be in flight; keep the operand pools alive and
        unmodified until their batch has ended."""
        B = A if B is None else B
        lhs, rhs = _pair_ids(lhs, rhs)
        rh = None
        if reuse is not None:
            rh, reuse.h = reuse.h, None  # consumed
        h = self.lib.rhip_pairwise_begin(self.h, OPS[op], A.h, B.h, lhs.size, lhs.ctypes.data, rhs.ctypes.data, rh)
        if not h:
            err = self._err()
            if reuse is not None and "still in flight" in err:
                reuse.h = rh  # refused, not consumed: the caller's Pool stays valid
            raise RoaringHipError(f"pairwise_begin {op} failed: " + err)
        return Batch(self, h, (A, B))

    # ---- prepared pair lists (rhip_pairlist_*): a list used more than once is validated / summed / uploaded once ----
    def pairlist(self, A: "Pool", lhs, B: Optional["Pool"] = None, rhs=None) -> "PairList":
        B = A if B is None else B
        lhs, rhs = _pair_ids(lhs, rhs)
        h = self.lib.rhip_pairlist_create(self.h, A.h, B.h, lhs.size, lhs.ctypes.data, rhs.ctypes.data)
        if not h:
            raise RoaringHipError("pairlist failed: " + self._err())
        return PairList(self, h, (A, B))

    def pairlist_all_pairs(self, A: "Pool") -> "PairList":
        """All unordered pairs (i < j) of one pool, row by row -- the reference benchmark's all-pairs loops."""
        h = self.lib.rhip_pairlist_all_pairs(self.h, A.h)
        if not h:
            raise RoaringHipError("pairlist_all_pairs failed: " + self._err())
        return PairList(self, h, (A, A))

    def pairlist_successive(self, A: "Pool") -> "PairList":
        """Successive bitmaps (i, i + 1) -- the reference benchmark's successive_* loops."""
        h = self.lib.rhip_pairlist_successive(self.h, A.h)
        if not h:
            raise RoaringHipError("pairlist_successive failed: " + self._err())
        return PairList(self, h, (A, A))

    def _ops_codes(self, ops):
        ops = [ops] if isinstance(ops, str) else list(ops)
        return ops, (C.c_int * len(ops))(*[OPS[o] for o in ops])

    def pairwise_list(self, ops, plist: "PairList", reuse: Optional["Pool"] = None) -> "Pool":
        """`pairwise` (ops = one name) or `pairwise_multi` (a list of names) over a prepared pair list."""
        return self.pairwise_list_begin(ops, plist, reuse).end()

    def pairwise_list_begin(self, ops, plist: "PairList", reuse: Optional["Pool"] = None) -> "Batch":
        ops, codes = self._ops_codes(ops)
        rh = None
        if reuse is not None:
            rh, reuse.h = reuse.h, None  # consumed
        h = self.lib.rhip_pairwise_list_begin(self.h, len(ops), codes, plist.h, rh)
        if not h:
            err = self._err()
            if reuse is not None and "still in flight" in err:
                reuse.h = rh
            raise RoaringHipError(f"pairwise_list_begin {ops} failed: " + err)
        return Batch(self, h, (plist,) + tuple(plist._operands))

    def pairwise_list_cardinality(self, op: str, plist: "PairList") -> np.ndarray:
        out = np.zeros(len(plist), dtype=np.uint64)
        if self.lib.rhip_pairwise_list_cardinality(self.h, OPS[op], plist.h, out.ctypes.data) != 0:
            raise RoaringHipError(f"pairwise_list_cardinality {op} failed: " + self._err())
        return out

    def pairwise_cardinality(self, op: str, A: "Pool", lhs, B: Optional["Pool"] = None, rhs=None) -> np.ndarray:
        """roaring_bitmap_{and,or,xor,andnot}_cardinality batched."""
        B = A if B is None else B
        lhs, rhs = _pair_ids(lhs, rhs)
        out = np.zeros(lhs.size, dtype=np.uint64)
        rc = self.lib.rhip_pairwise_cardinality(self.h, OPS[op], A.h, B.h, lhs.size, lhs.ctypes.data,
                                                rhs.ctypes.data, out.ctypes.data)
        if rc != 0:
            raise RoaringHipError(f"pairwise_cardinality {op} failed: " + self._err())
        return out

    def pairwise_predicate(self, pred: str, A: "Pool", lhs, B: Optional["Pool"] = None, rhs=None) -> np.ndarray:
        """roaring_bitmap_intersect / is_subset / is_strict_subset / equals batched; bool array."""
        B = A if B is None else B
        lhs, rhs = _pair_ids(lhs, rhs)
        out = np.zeros(lhs.size, dtype=np.uint8)
        rc = self.lib.rhip_pairwise_predicate(self.h, PREDS[pred], A.h, B.h, lhs.size, lhs.ctypes.data,
                                              rhs.ctypes.data, out.ctypes.data)
        if rc != 0:
            raise RoaringHipError(f"pairwise_predicate {pred} failed: " + self._err())
        return out.astype(bool)

    def pairwise_inplace(self, op: str, A: "Pool", lhs, B: Optional["Pool"] = None, rhs=None) -> None:
        """A[lhs[k]] <- op(A[lhs[k]], B[rhs[k]]) -- roaring_bitmap_*_inplace batched; lhs must not repeat."""
        B = A if B is None else B
        lhs, rhs = _pair_ids(lhs, rhs)
        rc = self.lib.rhip_pairwise_inplace(self.h, OPS[op], A.h, B.h, lhs.size, lhs.ctypes.data, rhs.ctypes.data)
        if rc != 0:
            raise RoaringHipError(f"pairwise_inplace {op} failed: " + self._err())

    # ---- reshaping -----------------------------------------------------
    def pool_select(self, pools: Sequence["Pool"], src_pool, src_bitmap) -> "Pool":
        """New pool whose bitmap i is pools[src_pool[i]][src_bitmap[i]] (device-side gather)."""
        sp, sb = _u32(src_pool), _u32(src_bitmap)
        if sp.shape != sb.shape:
            raise ValueError("src_pool/src_bitmap length mismatch")
        arr = (C.c_void_p * max(len(pools), 1))(*[p.h for p in pools])
        h = self.lib.rhip_pool_select(self.h, len(pools), arr, sp.size, sp.ctypes.data, sb.ctypes.data)
        if not h:
            raise RoaringHipError("pool_select failed: " + self._err())
        return Pool(self, h)

    def run_optimize(self, P: "Pool") -> "Pool":
        """roaring_bitmap_run_optimize applied to every bitmap of P (new pool)."""
        h = self.lib.rhip_pool_run_optimize(self.h, P.h)
        if not h:
            raise RoaringHipError("run_optimize failed: " + self._err())
        return Pool(self, h)

    def flip(self, P: "Pool", starts, ends) -> "Pool":
        """roaring_bitmap_flip(P[i], starts[i], ends[i]) for every bitmap (new pool)."""
        st = np.ascontiguousarray(starts, dtype=np.uint64)
        en = np.ascontiguousarray(ends, dtype=np.uint64)
        if st.size != len(P) or en.size != len(P):
            raise ValueError("one [start, end) range per bitmap")
        h = self.lib.rhip_pool_flip(self.h, P.h, st.ctypes.data, en.ctypes.data)
        if not h:
            raise RoaringHipError("flip failed: " + self._err())
        return Pool(self, h)

    def remove_run_compression(self, P: "Pool") -> "Pool":
        """roaring_bitmap_remove_run_compression applied to every bitmap of P (new pool)."""
        h = self.lib.rhip_pool_remove_run_compression(self.h, P.h)
        if not h:
            raise RoaringHipError("remove_run_compression failed: " + self._err())
        return Pool(self, h)

    # ---- many-way ------------------------------------------------------
    def _many(self, fn, P: "Pool", ids) -> "Pool":
        if ids is None:
            h = fn(self.h, P.h, 0, None)
        else:
            ids = _u32(ids)
            h = fn(self.h, P.h, ids.size, ids.ctypes.data)
        if not h:
            raise RoaringHipError("many-way aggregation failed: " + self._err())
        return Pool(self, h)

    def or_many(self, P: "Pool", ids=None) -> "Pool":
        """roaring_bitmap_or_many over P[ids] (whole pool if ids is None); one-bitmap pool."""
        return self._many(self.lib.rhip_or_many, P, ids)

    def xor_many(self, P: "Pool", ids=None) -> "Pool":
        return self._many(self.lib.rhip_xor_many, P, ids)

    def or_many_heap(self, P: "Pool", ids=None) -> "Pool":
        """roaring_bitmap_or_many_heap with the reference's container types: the size-ordered tournament itself (exact, n - 1
        dependent steps; or_many is the fast path for the same set)."""
        return self._many(self.lib.rhip_or_many_heap, P, ids)

    def many_partials(self, op: str, P: "Pool", ids=None) -> "PartialChunks":
        """Stage 1 of the sharded or_many/xor_many: one uncompressed 1024-word chunk per key."""
        out = Partials()
        if ids is None:
            rc = self.lib.rhip_many_partials(self.h, OPS[op], P.h, 0, None, C.byref(out))
        else:
            ids = _u32(ids)
            rc = self.lib.rhip_many_partials(self.h, OPS[op], P.h, ids.size, ids.ctypes.data, C.byref(out))
        if rc != 0:
            raise RoaringHipError("many_partials failed: " + self._err())
        return PartialChunks(self, out)

    def many_finalize(self, op: str, is64: bool, n_chunks: int, d_keys: int, d_words: int) -> "Pool":
        """Stage 2: combine (key, chunk) records by key and canonicalise; device pointers in."""
        h = self.lib.rhip_many_finalize(self.h, OPS[op], 1 if is64 else 0, n_chunks, d_keys, d_words)
        if not h:
            raise RoaringHipError("many_finalize failed: " + self._err())
        return Pool(self, h)

    def many_sharded_native(self, nccl_comm: int, op: str, P: "Pool", ids=None, key_space: int = 0) -> "Pool":
        """rhip_many_sharded: the sharded or_many / xor_many with the exchange done by the library itself over the
        caller's ncclComm_t (an address).  Returns this rank's share (keys with key % world == rank)."""
        out = C.c_void_p()
        if ids is None:
            rc = self.lib.rhip_many_sharded(self.h, nccl_comm, OPS[op], P.h, 0, None, int(key_space), C.byref(out))
        else:
            ids = _u32(ids)
            rc = self.lib.rhip_many_sharded(self.h, nccl_comm, OPS[op], P.h, ids.size, ids.ctypes.data, int(key_space), C.byref(out))
        if rc != 0 or not out.value:
            raise RoaringHipError("many_sharded failed: " + self._err())
        return Pool(self, out.value)

    def many_partials_dense(self, op: str, P: "Pool", ids, key_space: int, world: int, d_table: int) -> None:
        """Stage 1, dense exchange: ENQUEUES the reduction of P[ids] into the zero-filled table at device address
        d_table (world * ceil(key_space / world) rows of 1024 words; the chunk of key k at row (k % world) * B + k //
        world) and returns without waiting for the device."""
        if ids is None:
            rc = self.lib.rhip_many_partials_dense(self.h, OPS[op], P.h, 0, None, key_space, world, d_table)
        else:
            ids = _u32(ids)
            rc = self.lib.rhip_many_partials_dense(self.h, OPS[op], P.h, ids.size, ids.ctypes.data, key_space, world, d_table)
        if rc != 0:
            raise RoaringHipError("many_partials_dense failed: " + self._err())

    def many_finalize_dense(self, op: str, is64: bool, world: int, rank: int, keys_per_rank: int, d_table: int) -> "Pool":
        """Stage 3, dense exchange: row s * keys_per_rank + j of the received table = source rank s's chunk of key
        rank + world * j.  The one host wait of the sharded pipeline."""
        h = self.lib.rhip_many_finalize_dense(self.h, OPS[op], 1 if is64 else 0, world, rank, keys_per_rank, d_table)
        if not h:
            raise RoaringHipError("many_finalize_dense failed: " + self._err())
        return Pool(self, h)


    # ---- torch interop (only croaring_amd.distributed uses these; torch is imported lazily) ---------------
    def torch_device(self):
        import torch
        return torch.device("cuda", torch.cuda.current_device())

    def torch_stream(self):
        """Context manager making the engine's HIP stream torch's current stream: torch work issued inside is
        ordered with the engine's kernels by the stream itself, no host synchronisation needed."""
        import torch
        return torch.cuda.stream(torch.cuda.ExternalStream(self.stream, device=self.torch_device()))

    def as_tensor(self, ptr: int, shape):
        """Zero-copy int64 torch view of engine-owned device memory."""
        import torch

        class _DevArray:
            def __init__(self, ptr, shape):
                self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<i8", "data": (ptr, False),
                                                 "version": 2, "strides": None}
        return torch.as_tensor(_DevArray(ptr, shape), device=self.torch_device())


class PartialChunks:
    def __init__(self, eng: Engine, p: Partials):
        self.eng, self.p = eng, p

    n_keys = property(lambda self: int(self.p.n_keys))
    d_keys = property(lambda self: int(self.p.d_keys or 0))
    d_words = property(lambda self: int(self.p.d_words or 0))
    max_key = property(lambda self: int(self.p.max_key))

    def free(self):
        if self.p is not None:
            self.eng.lib.rhip_partials_free(self.eng.h, C.byref(self.p))
            self.p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PairList:
    """A prepared pair list (rhip_pairlist_t); keeps its operand pools alive."""

    def __init__(self, engine: "Engine", handle, operands):
        self.engine, self.h, self._operands = engine, handle, operands

    def __len__(self) -> int:
        return int(self.engine.lib.rhip_pairlist_size(self.h))

    def pairs(self):
        n = len(self)
        lhs, rhs = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)
        if self.engine.lib.rhip_pairlist_pairs(self.h, lhs.ctypes.data, rhs.ctypes.data) != 0:
            raise RoaringHipError(self.engine._err())
        return lhs, rhs

    def drop_plans(self) -> int:
        """Forget the plans cached with the list (rhip_pairlist_drop_plans); the next batch of each op plans afresh."""
        return int(self.engine.lib.rhip_pairlist_drop_plans(self.h)) if self.h else 0

    def free(self):
        if getattr(self, "h", None) and self.engine is not None and self.engine.h:
            self.engine.lib.rhip_pairlist_free(self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Batch:
    """A pairwise batch in flight (rhip_batch_t): `end()` waits for it and returns the result Pool."""

    def __init__(self, engine: "Engine", handle, operands):
        self.engine, self.h, self._operands = engine, handle, operands  # operands kept alive until the batch ends

    def end(self) -> "Pool":
        if self.h is None:
            raise RoaringHipError("batch already ended")
        h, self.h = self.h, None
        r = self.engine.lib.rhip_pairwise_end(h)
        self._operands = None
        if not r:
            raise RoaringHipError("pairwise_end failed: " + self.engine._err())
        return Pool(self.engine, r)

    def __del__(self):  # an abandoned batch still has to be ended: its slot and its result pool belong to it
        try:
            if self.h is not None and self.engine is not None and self.engine.h:
                r = self.engine.lib.rhip_pairwise_end(self.h)
                self.h = None
                if r:
                    self.engine.lib.rhip_pool_free(r)
        except Exception:
            pass


class Pool:
    """A device-resident set of bitmaps (rhip_pool_t)."""

    def __init__(self, eng: Engine, h):
        self.eng, self.h = eng, h

    def free(self):
        if getattr(self, "h", None):
            self.eng.lib.rhip_pool_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def __len__(self) -> int:
        return self.eng.lib.rhip_pool_size(self.h)

    @property
    def n_containers(self) -> int:
        return self.eng.lib.rhip_pool_containers(self.h)

    @property
    def is64(self) -> bool:
        return bool(self.eng.lib.rhip_pool_is64(self.h))

    def payload_bytes(self) -> int:
        return self.eng.lib.rhip_pool_payload_bytes(self.h)

    def arena_bytes(self) -> int:
        """HBM bytes of the payload arena, padding included (payload_bytes() is the algorithmic figure)."""
        return self.eng.lib.rhip_pool_arena_bytes(self.h)

    @property
    def payload_align(self) -> int:
        """Slot granule the loader chose: 16, or 128 (whole cache lines; RHIP_POOL_ALIGN pins it)."""
        return self.eng.lib.rhip_pool_payload_align(self.h)

    def type_counts(self) -> tuple:
        out = (C.c_uint64 * 3)()
        if self.eng.lib.rhip_pool_type_counts(self.h, out) != 0:
            raise RoaringHipError(self.eng._err())
        return tuple(int(x) for x in out)  # (bitset, array, run)

    def max_key(self) -> int:
        """Largest container key of the pool (cached on the pool after the first call)."""
        out = C.c_uint64(0)
        if self.eng.lib.rhip_pool_max_key(self.h, C.byref(out)) != 0:
            raise RoaringHipError("pool_max_key failed: " + self.eng._err())
        return int(out.value)

    def cardinalities(self) -> np.ndarray:
        """roaring_bitmap_get_cardinality of every bitmap."""
        out = np.zeros(len(self), dtype=np.uint64)
        if self.eng.lib.rhip_pool_cardinalities(self.h, out.ctypes.data) != 0:
            raise RoaringHipError(self.eng._err())
        return out

    def serialize(self, i: int) -> bytes:
        """roaring_bitmap_portable_serialize of bitmap i."""
        n = self.eng.lib.rhip_pool_portable_size(self.h, i)
        if n == 0:
            raise RoaringHipError("portable_size failed: " + self.eng._err())
        buf = C.create_string_buffer(n)
        w = self.eng.lib.rhip_pool_portable_serialize(self.h, i, buf)
        if w != n:
            raise RoaringHipError(f"serialize wrote {w} of {n} bytes: " + self.eng._err())
        return buf.raw

    def to_values(self):
        """roaring_bitmap_to_uint32_array (to_uint64_array for 64-bit pools) of every bitmap, decoded on the device:
        (values, offsets[n+1]); bitmap i = values[offsets[i]:offsets[i+1]]."""
        n = len(self)
        offs = np.zeros(n + 1, dtype=np.uint64)
        fn = self.eng.lib.rhip_pool_to_u64 if self.is64 else self.eng.lib.rhip_pool_to_u32
        if fn(self.h, None, 0, offs.ctypes.data) != 0:
            raise RoaringHipError("to_values failed: " + self.eng._err())
        vals = np.empty(int(offs[n]), dtype=np.uint64 if self.is64 else np.uint32)
        if vals.size and fn(self.h, vals.ctypes.data, vals.size, None) != 0:
            raise RoaringHipError("to_values failed: " + self.eng._err())
        return vals, offs

    def serialize_many(self, ids=None):
        """Portable images of bitmaps `ids` (None: all) packed back to back: (uint8 blob, uint64 offsets[n+1]),
        assembled on the device, one download (64-bit pools: whole pool only)."""
        ids_a = None if ids is None else _u32(ids)
        n = len(self) if ids_a is None else ids_a.size
        ip = None if ids_a is None else ids_a.ctypes.data
        offs = np.zeros(n + 1, dtype=np.uint64)
        if self.eng.lib.rhip_pool_portable_sizes(self.h, n, ip, offs.ctypes.data) != 0:
            raise RoaringHipError("portable_sizes failed: " + self.eng._err())
        blob = np.empty(int(offs[n]), dtype=np.uint8)
        if n and int(offs[n]):
            w = self.eng.lib.rhip_pool_portable_serialize_many(self.h, n, ip, blob.ctypes.data, blob.size, None)
            if w != blob.size:
                raise RoaringHipError(f"serialize_many wrote {w} of {blob.size} bytes: " + self.eng._err())
        return blob, offs

    def frozen_serialize_many(self, ids=None):
        """Frozen images of bitmaps `ids` (None: all), packed at 32-byte aligned offsets: (uint8 blob, uint64
        offsets[n+1], uint64 lens[n]); image k = blob[offsets[k] : offsets[k] + lens[k]], byte-identical to
        roaring_bitmap_frozen_serialize.  32-bit pools only."""
        ids_a = None if ids is None else _u32(ids)
        n = len(self) if ids_a is None else ids_a.size
        ip = None if ids_a is None else ids_a.ctypes.data
        offs = np.zeros(n + 1, dtype=np.uint64)
        lens = np.zeros(max(n, 1), dtype=np.uint64)
        if self.eng.lib.rhip_pool_frozen_sizes(self.h, n, ip, offs.ctypes.data, lens.ctypes.data) != 0:
            raise RoaringHipError("frozen_sizes failed: " + self.eng._err())
        blob = np.empty(int(offs[n]), dtype=np.uint8)
        if n and int(offs[n]):
            w = self.eng.lib.rhip_pool_frozen_serialize_many(self.h, n, ip, blob.ctypes.data, blob.size, None, None)
            if w != blob.size:
                raise RoaringHipError(f"frozen_serialize_many wrote {w} of {blob.size} bytes: " + self.eng._err())
        return blob, offs, lens[:n]

    def frozen_serialize_all(self) -> list:
        blob, offs, lens = self.frozen_serialize_many()
        raw = blob.tobytes()
        return [raw[int(offs[i]):int(offs[i]) + int(lens[i])] for i in range(len(self))]

    def serialize_all(self) -> list:
        blob, offs = self.serialize_many()
        raw = blob.tobytes()
        return [raw[int(offs[i]):int(offs[i + 1])] for i in range(len(self))]
