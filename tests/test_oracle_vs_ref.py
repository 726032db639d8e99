"""Randomised differential test: oracle restatement vs the REAL CRoaring (oracle/_ref), byte level.
Skipped where the reference library was not prebuilt (it cannot be built without /root/reference)."""
import numpy as np
import pytest

from gen_inputs import random_bitmap
from util import OPS


def test_pairwise_bytes(oracle, ref):
    rng = np.random.default_rng(42)
    for it in range(150):
        va, vb = random_bitmap(rng), random_bitmap(rng)
        ra, rb = ref.from_sorted(va), ref.from_sorted(vb)
        sa, sb = ref.serialize(ra), ref.serialize(rb)
        oa, ob = oracle.deserialize(sa), oracle.deserialize(sb)
        oa2 = oracle.from_sorted(va)
        assert oracle.serialize(oa2) == sa, "run_optimize parity"
        for op in OPS:
            rr, oo = ref.op(op, ra, rb), oracle.op(op, oa, ob)
            assert ref.serialize(rr) == oracle.serialize(oo), (it, op)
            assert ref.op_cardinality(op, ra, rb) == oracle.op_cardinality(op, oa, ob) == ref.cardinality(rr)
            assert oracle.validate(oo) and ref.validate(rr)
            ref.free(rr)
            oracle.free(oo)
        for h in (ra, rb):
            ref.free(h)
        for h in (oa, ob, oa2):
            oracle.free(h)


def test_many_way(oracle, ref):
    rng = np.random.default_rng(43)
    for it in range(60):
        n = int(rng.integers(0, 9))
        vs = [random_bitmap(rng, max_keys=6, key_space=8) for _ in range(n)]
        rs = [ref.from_sorted(v) for v in vs]
        os_ = [oracle.deserialize(ref.serialize(r)) for r in rs]
        rr, oo = ref.or_many(rs), oracle.or_many(os_)
        assert ref.serialize(rr) == oracle.serialize(oo), f"or_many {it}"
        rx, ox = ref.xor_many(rs), oracle.xor_many(os_)
        assert ref.serialize(rx) == oracle.serialize(ox), f"xor_many {it}"
        rh, oh = ref.or_many_heap(rs), oracle.or_many_heap(os_)
        assert ref.serialize(rh) == oracle.serialize(oh), f"or_many_heap {it}"
        for h in rs + [rr, rx, rh]:
            ref.free(h)
        for h in os_ + [oo, ox, oh]:
            oracle.free(h)


def test_xor_many_fold_typing(oracle, ref):
    """roaring_bitmap_xor_many is a fixed left fold (roaring.c:795-809): lazy_xor, lazy_xor_inplace ..., repair.  Its
    container TYPES follow from the fold (a run accumulator survives R ^ R and R ^ small array, an accumulator that
    empties is removed and re-cloned ...): bytes, on few keys with many members, run-heavy, with and without run
    compression of the inputs, including duplicated members (which cancel)."""
    rng = np.random.default_rng(4343)
    mixes = (("runs", "shortruns", "tiny", "single", "edge"), ("runs", "tiny"), ("runs", "shortruns", "dense", "sparse"),
             ("sparse", "tiny", "mid", "boundary4096"), ("full", "nearfull", "runs", "blocks", "verydense"), None)
    for it in range(240):
        profs = mixes[it % len(mixes)]
        n = int(rng.integers(2, 12))
        kw = dict(max_keys=3, key_space=3) if profs is None else dict(max_keys=3, key_space=3, profiles=profs)
        vs = [random_bitmap(rng, **kw) for _ in range(n)]
        if it % 5 == 0 and n > 3:  # a member twice: the accumulator passes through the empty set
            vs[2] = vs[0]; vs[1] = vs[0] if it % 10 == 0 else vs[1]
        rs = [ref.from_sorted(v) for v in vs]
        if it % 3 == 1:
            for r in rs[::2]:
                ref.remove_run_compression(r)
        os_ = [oracle.deserialize(ref.serialize(r)) for r in rs]
        rx, ox = ref.xor_many(rs), oracle.xor_many(os_)
        assert ref.serialize(rx) == oracle.serialize(ox), f"xor_many {it}"
        assert oracle.validate(ox)
        for h in rs + [rx]:
            ref.free(h)
        for h in os_ + [ox]:
            oracle.free(h)


def test_or_many_heap_tournament_typing(oracle, ref):
    """roaring_bitmap_or_many_heap (roaring_priority_queue.c:200-247): the tournament ordered by serialized size, lazy
    unions without early bitset conversion (arrays stay arrays up to 1024 values, run | array stays a raw run, run | run is
    typed by size at every step, a bitset absorbs, full runs short-circuit), one repair pass -- BYTES, on few keys with many
    members, equal-sized inputs (ties are broken by heap position), empty bitmaps, with and without run compression."""
    rng = np.random.default_rng(5151)
    mixes = (("runs", "shortruns", "tiny", "single", "edge"), ("runs", "tiny"), ("runs", "shortruns", "dense", "sparse"),
             ("sparse", "tiny", "mid", "boundary4096"), ("full", "nearfull", "runs", "blocks", "verydense"), None)
    for it in range(300):
        profs = mixes[it % len(mixes)]
        n = int(rng.integers(2, 24))
        kw = dict(max_keys=4, key_space=4) if profs is None else dict(max_keys=4, key_space=4, profiles=profs)
        vs = [random_bitmap(rng, **kw) for _ in range(n)]
        if it % 7 == 0:
            vs[int(rng.integers(0, n))] = np.zeros(0, np.uint32)
        if it % 5 == 0 and n > 3:  # equal sizes: the same bitmap several times
            vs[2] = vs[0]; vs[3] = vs[0]
        rs = [ref.from_sorted(v) for v in vs]
        if it % 3 == 1:
            for r in rs[::2]:
                ref.remove_run_compression(r)
        os_ = [oracle.deserialize(ref.serialize(r)) for r in rs]
        rh, oh = ref.or_many_heap(rs), oracle.or_many_heap(os_)
        assert ref.serialize(rh) == oracle.serialize(oh), f"or_many_heap {it}"
        for h in rs + [rh]:
            ref.free(h)
        for h in os_ + [oh]:
            oracle.free(h)


def test_64bit(oracle, ref):
    rng = np.random.default_rng(44)
    for it in range(40):
        def mk():
            parts = [(np.uint64(int(hi)) << np.uint64(32)) | random_bitmap(rng, max_keys=4, key_space=6).astype(np.uint64)
                     for hi in rng.choice(5, int(rng.integers(0, 4)), replace=False)]
            return np.sort(np.concatenate(parts)) if parts else np.zeros(0, np.uint64)
        va, vb = mk(), mk()
        ra, rb = ref.from_sorted64(va), ref.from_sorted64(vb)
        oa, ob = oracle.deserialize64(ref.serialize64(ra)), oracle.deserialize64(ref.serialize64(rb))
        assert oracle.serialize64(oa) == ref.serialize64(ra)
        for op in OPS:
            rr, oo = ref.op64(op, ra, rb), oracle.op64(op, oa, ob)
            assert ref.serialize64(rr) == oracle.serialize64(oo), (it, op)
            assert ref.cardinality64(rr) == oracle.cardinality64(oo)
            ref.free64(rr)
            oracle.free64(oo)
        for h in (ra, rb):
            ref.free64(h)
        for h in (oa, ob):
            oracle.free64(h)


def test_or_many_full_container_orderings(oracle, ref):
    """Every ordering of an adversarial set whose union fills the chunk: the run-vs-bitset outcome of
    roaring_bitmap_or_many depends on the fold order (roaring.c:2529-2548 vs 2619-2647)."""
    import itertools
    full = np.arange(65536, dtype=np.uint32)
    sets = {"fullrun": (full, True), "fullbits": (full, False),
            "evens": (np.arange(0, 65536, 2, dtype=np.uint32), False),
            "odds": (np.arange(1, 65536, 2, dtype=np.uint32), False),
            "lo": (np.arange(0, 40000, dtype=np.uint32), True), "hi": (np.arange(30000, 65536, dtype=np.uint32), True),
            "few": (np.array([5, 77, 4000], dtype=np.uint32), True)}
    names = list(sets)
    rs = {n: ref.from_sorted(v, run_optimize=ro) for n, (v, ro) in sets.items()}
    os_ = {n: oracle.deserialize(ref.serialize(rs[n])) for n in names}
    for r in (2, 3, 4):
        for combo in itertools.permutations(range(len(names)), r):
            a = ref.or_many([rs[names[i]] for i in combo])
            b = oracle.or_many([os_[names[i]] for i in combo])
            assert ref.serialize(a) == oracle.serialize(b), [names[i] for i in combo]
            ref.free(a)
            oracle.free(b)


def test_conversions_and_predicates(oracle, ref):
    """oc_run_optimize / oc_remove_run_compression / oc_intersect / oc_is_subset / oc_is_strict_subset / oc_equals
    vs roaring_bitmap_run_optimize (roaring.c:1530), _remove_run_compression (:1564), _intersect (:2998),
    _is_subset (:2151), _is_strict_subset (:3172), _equals (:2128)."""
    from gen_inputs import random_bitmap
    rng = np.random.default_rng(31)
    vals = [random_bitmap(rng, max_keys=6, key_space=8) for _ in range(80)]
    oh = [oracle.from_sorted(v, run_optimize=bool(i & 1)) for i, v in enumerate(vals)]
    rh = [ref.from_sorted(v, run_optimize=bool(i & 1)) for i, v in enumerate(vals)]
    for a, b in zip(oh, rh):
        assert oracle.serialize(a) == ref.serialize(b)
    # engineered subsets
    for i in range(0, 40, 2):
        oh.append(oracle.op("and", oh[i], oh[i + 1]))
        rh.append(ref.op("and", rh[i], rh[i + 1]))
    n = len(oh)
    pairs = [(int(a), int(b)) for a, b in zip(rng.integers(0, n, 500), rng.integers(0, n, 500))]
    pairs += [(80 + k, 2 * k) for k in range(20)] + [(k, k) for k in range(10)]
    for pred in ("intersect", "is_subset", "is_strict_subset", "equals"):
        for a, b in pairs:
            assert oracle.predicate(pred, oh[a], oh[b]) == ref.predicate(pred, rh[a], rh[b]), (pred, a, b)
    for mode in ("run_optimize", "remove_run_compression"):
        for a, b in zip(oh, rh):
            ca, cb = oracle.deserialize(oracle.serialize(a)), ref.deserialize(ref.serialize(b))
            assert getattr(oracle, mode)(ca) == getattr(ref, mode)(cb), mode
            assert oracle.serialize(ca) == ref.serialize(cb), mode
            assert oracle.validate(ca)
            oracle.free(ca)
            ref.free(cb)
    for h in oh:
        oracle.free(h)
    for h in rh:
        ref.free(h)


def _flip_ranges(rng, n):
    """Ranges that hit every branch of roaring_bitmap_flip: inside one container, container-aligned, spanning many
    keys, touching 0 / 2^32, empty, reversed, beyond 32 bits."""
    out = [(0, 1 << 32), (0, 0), (5, 5), (9, 3), (0, 65536), (65536, 131072), (1, 65535), (65535, 65537),
           ((1 << 32) - 1, 1 << 32), (1 << 32, (1 << 32) + 10), ((1 << 32) + 1, (1 << 32) + 9), (100, (1 << 32) + 4),
           (3 << 16, (9 << 16) + 17), ((3 << 16) + 5, 9 << 16)]
    for _ in range(n):
        a = int(rng.integers(0, 1 << 21))
        b = a + int(rng.choice([1, 2, 3, 100, 5000, 65536, 70000, 300000, 1 << 20]))
        out.append((a, b))
    return out


def test_flip(oracle, ref):
    """oc_flip vs roaring_bitmap_flip (roaring.c:2289-2342), byte level."""
    from gen_inputs import random_bitmap
    rng = np.random.default_rng(33)
    vals = [random_bitmap(rng, max_keys=8, key_space=20) for _ in range(40)] + [np.zeros(0, np.uint32)]
    for i, v in enumerate(vals):
        a, b = oracle.from_sorted(v, run_optimize=bool(i & 1)), ref.from_sorted(v, run_optimize=bool(i & 1))
        for s, e in _flip_ranges(rng, 12):
            fa, fb = oracle.flip(a, s, e), ref.flip(b, s, e)
            assert oracle.serialize(fa) == ref.serialize(fb), (i, s, e)
            assert oracle.validate(fa)
            oracle.free(fa)
            ref.free(fb)
        oracle.free(a)
        ref.free(b)


def test_flip_64bit_oracle_vs_ref(oracle, ref):
    """oc64_flip == the real roaring64_bitmap_flip, byte level, on the cases the GPU test uses."""
    from test_gpu_poolops import _flip64_cases
    rng = np.random.default_rng(6464)
    for i, (v, lo, hi) in enumerate(_flip64_cases(rng)):
        ho, hr = oracle.from_sorted64(v, run_optimize=bool(i & 1)), ref.from_sorted64(v, run_optimize=bool(i & 1))
        assert oracle.serialize64(ho) == ref.serialize64(hr)
        fo, fr = oracle.flip64(ho, lo, hi), ref.flip64(hr, lo, hi)
        assert oracle.serialize64(fo) == ref.serialize64(fr), (i, lo, hi)
        for h in (ho, fo):
            oracle.free64(h)
        for h in (hr, fr):
            ref.free64(h)


def test_frozen_format(oracle, ref):
    """roaring_bitmap_frozen_size_in_bytes / _serialize / _view (src/roaring.c:3207-3457) against the restatement on
    random bitmaps: identical images, each side reads the other's, same acceptance of damaged images."""
    from gen_inputs import random_bitmap
    rng = np.random.default_rng(77)
    for t in range(150):
        v = random_bitmap(rng)
        ho, hr = oracle.from_sorted(v), ref.from_sorted(v)
        fo, fr = oracle.frozen_serialize(ho), ref.frozen_serialize(hr)
        assert fo == fr, t
        h2 = oracle.frozen_deserialize(fr)
        h3 = ref.frozen_deserialize(fo)
        assert h2 and h3 and oracle.serialize(h2) == ref.serialize(h3) == ref.serialize(hr), t
        for cut in (1, 2, 4, 7):
            d = fr[:-cut] if len(fr) > cut else b""
            assert (oracle.frozen_deserialize(d) is None) == (ref.frozen_deserialize(d) is None), (t, cut)
        if len(fr) > 8:
            d = bytearray(fr)
            d[int(rng.integers(max(0, len(d) - 64), len(d)))] ^= 1 << int(rng.integers(0, 8))
            a, b = oracle.frozen_deserialize(bytes(d)), ref.frozen_deserialize(bytes(d))
            assert (a is None) == (b is None), t
            if a:
                oracle.free(a)
            if b:
                ref.free(b)
        oracle.free(ho); oracle.free(h2); ref.free(hr); ref.free(h3)
