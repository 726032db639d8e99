"""GPU parity tests of the device-resident reshaping entry points: rhip_pool_select,
rhip_pairwise_inplace, rhip_pool_run_optimize / _remove_run_compression, rhip_pairwise_predicate -- through the
C ABI, against the oracle (and the real reference where oracle/_ref is present), byte level."""
import struct

import numpy as np
import pytest

from gen_inputs import PROFILES, chunk_values, random_bitmap
from util import OPS

pytestmark = pytest.mark.gpu


def _mixed_pool(oracle, rng, n=60, run_optimize=True):
    hs = [oracle.from_sorted(random_bitmap(rng, max_keys=8, key_space=12), run_optimize=run_optimize) for _ in range(n)]
    return hs, [oracle.serialize(h) for h in hs]


def test_pool_select(engine, oracle):
    rng = np.random.default_rng(21)
    hs, bufs = _mixed_pool(oracle, rng, 40)
    A = engine.pool_from_serialized(bufs)
    lhs = rng.integers(0, 40, 30).astype(np.uint32)
    rhs = rng.integers(0, 40, 30).astype(np.uint32)
    R = engine.pairwise("xor", A, lhs, A, rhs)          # a result pool: slots larger than payloads, holes
    want_r = [R.serialize(k) for k in range(30)]
    sp = rng.integers(0, 2, 100).astype(np.uint32)
    sb = np.where(sp == 0, rng.integers(0, 40, 100), rng.integers(0, 30, 100)).astype(np.uint32)
    S = engine.pool_select([A, R], sp, sb)
    assert len(S) == 100 and not S.is64
    for i in range(100):
        assert S.serialize(i) == (bufs[sb[i]] if sp[i] == 0 else want_r[sb[i]]), i
    ca, cr = A.cardinalities(), R.cardinalities()
    assert np.array_equal(S.cardinalities(), np.array([ca[b] if p == 0 else cr[b] for p, b in zip(sp, sb)], np.uint64))
    # a selected pool is an ordinary operand
    T = engine.pairwise("or", S, np.arange(50, dtype=np.uint32), S, np.arange(50, 100, dtype=np.uint32))
    for k in (0, 7, 49):
        oa, ob = oracle.deserialize(S.serialize(k)), oracle.deserialize(S.serialize(50 + k))
        oo = oracle.op("or", oa, ob)
        assert T.serialize(k) == oracle.serialize(oo)
        for h in (oa, ob, oo):
            oracle.free(h)
    # empty selection, selection of empty bitmaps, argument errors
    assert len(engine.pool_select([A], [], [])) == 0
    e = oracle.from_sorted(np.zeros(0, np.uint32))
    E = engine.pool_from_serialized([oracle.serialize(e)])
    Z = engine.pool_select([E, A], [0, 1, 0], [0, 3, 0])
    assert Z.serialize(0) == oracle.serialize(e) and Z.serialize(1) == bufs[3] and Z.serialize(2) == oracle.serialize(e)
    with pytest.raises(Exception):
        engine.pool_select([A], [0], [40])
    with pytest.raises(Exception):
        engine.pool_select([A], [1], [0])
    for h in hs + [e]:
        oracle.free(h)


def test_pool_select_64bit(engine, oracle):
    rng = np.random.default_rng(22)
    vals = [np.unique(rng.integers(0, 1 << 34, 3000).astype(np.uint64)) for _ in range(6)]
    hs = [oracle.from_sorted64(v) for v in vals]
    bufs = [oracle.serialize64(h) for h in hs]
    P = engine.pool_from_serialized64(bufs)
    S = engine.pool_select([P], [0, 0, 0], [5, 0, 5])
    assert S.is64 and S.serialize(0) == bufs[5] and S.serialize(1) == bufs[0] and S.serialize(2) == bufs[5]
    a32 = oracle.from_sorted(np.arange(10, dtype=np.uint32))
    Q = engine.pool_from_serialized([oracle.serialize(a32)])
    with pytest.raises(Exception):
        engine.pool_select([P, Q], [0, 1], [0, 0])          # mixed key widths
    for h in hs:
        oracle.free64(h)
    oracle.free(a32)


@pytest.mark.parametrize("op", OPS)
def test_pairwise_inplace(engine, oracle, op):
    rng = np.random.default_rng(23)
    hs, bufs = _mixed_pool(oracle, rng, 50)
    hb, bufs_b = _mixed_pool(oracle, rng, 20)
    A = engine.pool_from_serialized(bufs)
    B = engine.pool_from_serialized(bufs_b)
    lhs = rng.choice(50, 17, replace=False).astype(np.uint32)
    rhs = rng.integers(0, 20, 17).astype(np.uint32)
    engine.pairwise_inplace(op, A, lhs, B, rhs)
    assert len(A) == 50
    touched = {int(l): int(r) for l, r in zip(lhs, rhs)}
    cards = A.cardinalities()
    for i in range(50):
        if i in touched:
            oo = oracle.op(op, hs[i], hb[touched[i]])
            assert A.serialize(i) == oracle.serialize(oo), (op, i)
            assert cards[i] == oracle.cardinality(oo)
            oracle.free(oo)
        else:
            assert A.serialize(i) == bufs[i], (op, i, "untouched bitmap changed")
    # B may be A itself: a[i] op= a[j] reads the pre-update a[j]
    A2 = engine.pool_from_serialized(bufs)
    l2 = np.array([0, 1, 2, 3], np.uint32)
    r2 = np.array([1, 0, 3, 3], np.uint32)
    engine.pairwise_inplace(op, A2, l2, A2, r2)
    for l, r in zip(l2, r2):
        oo = oracle.op(op, hs[l], hs[r])
        assert A2.serialize(int(l)) == oracle.serialize(oo), (op, "self", int(l))
        oracle.free(oo)
    assert A2.serialize(4) == bufs[4]
    with pytest.raises(Exception):
        engine.pairwise_inplace(op, A2, [5, 5], B, [0, 1])    # repeated target
    with pytest.raises(Exception):
        engine.pairwise_inplace(op, A2, [50], B, [0])
    engine.pairwise_inplace(op, A2, [], B, [])                # empty batch is a no-op
    assert A2.serialize(4) == bufs[4]
    for h in hs + hb:
        oracle.free(h)


def _inefficient_run_bitmap():
    """One run container with 3000 runs of 10 values (12000-byte payload): valid, not run-efficient."""
    starts = np.arange(3000, dtype=np.uint32) * 20
    runs = np.stack([starts, np.full(3000, 9, np.uint32)], 1).astype(np.uint16)
    return struct.pack("<I", 12347 | (0 << 16)) + b"\x01" + struct.pack("<HH", 3, 3000 * 10 - 1) + \
        struct.pack("<H", 3000) + runs.tobytes()


def _conversion_inputs(oracle, rng):
    """Un-optimised bitmaps of every profile (arrays / bitsets that should become runs and ones that should not),
    optimised ones (runs that stay), hand-made inefficient runs."""
    hs = []
    for prof in PROFILES:
        for _ in range(3):
            keys = np.sort(rng.choice(50, 3, replace=False)).astype(np.uint32)
            v = np.concatenate([(k << np.uint32(16)) | chunk_values(rng, prof).astype(np.uint32) for k in keys])
            hs.append(oracle.from_sorted(v, run_optimize=False))
            hs.append(oracle.from_sorted(v, run_optimize=True))
    for _ in range(30):
        hs.append(oracle.from_sorted(random_bitmap(rng, max_keys=10, key_space=14), run_optimize=bool(rng.integers(0, 2))))
    hs.append(oracle.deserialize(_inefficient_run_bitmap()))
    hs.append(oracle.from_sorted(np.zeros(0, np.uint32)))
    # boundary: arrays / bitsets right at the size tie (2 + 4 n_runs == 2 card, == 8192)
    hs.append(oracle.from_sorted(np.concatenate([np.arange(0, 9), [100]]).astype(np.uint32), run_optimize=False))
    tie = np.concatenate([np.arange(s, s + 2) for s in range(0, 4 * 2047, 4)] + [np.arange(20000, 30000)])
    hs.append(oracle.from_sorted(tie.astype(np.uint32), run_optimize=False))
    return hs


@pytest.mark.parametrize("mode", ["run_optimize", "remove_run_compression"])
def test_container_conversions(engine, oracle, mode):
    rng = np.random.default_rng(24)
    hs = _conversion_inputs(oracle, rng)
    bufs = [oracle.serialize(h) for h in hs]
    P = engine.pool_from_serialized(bufs)
    Q = getattr(engine, mode)(P)
    assert len(Q) == len(P) and Q.n_containers == P.n_containers
    assert np.array_equal(Q.cardinalities(), P.cardinalities())
    bad = []
    for i, h in enumerate(hs):
        getattr(oracle, mode)(h)            # the oracle converts in place
        if Q.serialize(i) != oracle.serialize(h):
            bad.append(i)
    assert not bad, f"{mode}: {len(bad)} bitmaps differ, first {bad[:8]}"
    if mode == "remove_run_compression":
        assert Q.type_counts()[2] == 0
    # idempotent, and the converted pool is an ordinary operand
    Q2 = getattr(engine, mode)(Q)
    for i in range(0, len(hs), 7):
        assert Q2.serialize(i) == Q.serialize(i)
    R = engine.pairwise("and", Q, [0, 1, 2], P, [0, 1, 2])
    for k in range(3):
        assert R.serialize(k) is not None and R.cardinalities()[k] == P.cardinalities()[k]
    # results of set operations are not always storage-optimal (e.g. array x array unions): run_optimize on a
    # result pool equals the reference's op followed by roaring_bitmap_run_optimize
    n = len(hs)
    lhs = rng.integers(0, n, 80).astype(np.uint32)
    rhs = rng.integers(0, n, 80).astype(np.uint32)
    for op in ("or", "andnot"):
        res = getattr(engine, mode)(engine.pairwise(op, Q, lhs, Q, rhs))
        for k in range(80):
            oo = oracle.op(op, hs[lhs[k]], hs[rhs[k]])
            getattr(oracle, mode)(oo)
            assert res.serialize(k) == oracle.serialize(oo), (mode, op, k)
            oracle.free(oo)
    for h in hs:
        oracle.free(h)


def test_container_conversions_vs_reference(engine, ref):
    """Same check against the real CRoaring (oracle/_ref) where it is present."""
    rng = np.random.default_rng(25)
    hs = [ref.from_sorted(random_bitmap(rng, max_keys=10, key_space=14), run_optimize=bool(i & 1)) for i in range(60)]
    bufs = [ref.serialize(h) for h in hs]
    P = engine.pool_from_serialized(bufs)
    for mode in ("run_optimize", "remove_run_compression"):
        Q = getattr(engine, mode)(P)
        for i, h in enumerate(hs):
            g = ref.deserialize(bufs[i])
            getattr(ref, mode)(g)
            assert Q.serialize(i) == ref.serialize(g), (mode, i)
            ref.free(g)
    for h in hs:
        ref.free(h)


def test_pairwise_predicates(engine, oracle):
    rng = np.random.default_rng(26)
    hs, bufs = _mixed_pool(oracle, rng, 40)
    # engineered relations: subsets (and, andnot of a member), equal copies in another representation, empties
    extra = []
    for i in range(10):
        a, b = hs[i], hs[i + 10]
        extra += [oracle.op("and", a, b), oracle.op("andnot", a, b), oracle.op("or", a, b)]
    same = [oracle.deserialize(bufs[i]) for i in range(5)]
    for h in same:
        oracle.remove_run_compression(h)     # same set, different container types
    empty = oracle.from_sorted(np.zeros(0, np.uint32))
    allh = hs + extra + same + [empty]
    P = engine.pool_from_serialized([oracle.serialize(h) for h in allh])
    n = len(allh)
    lhs = np.concatenate([rng.integers(0, n, 400), np.arange(40, 70), np.arange(70, 75), [n - 1, n - 1, 0]]).astype(np.uint32)
    rhs = np.concatenate([rng.integers(0, n, 400), np.repeat(np.arange(10), 3), np.arange(0, 5), [n - 1, 3, n - 1]]).astype(np.uint32)
    seen = {}
    for pred in ("intersect", "is_subset", "is_strict_subset", "equals"):
        got = engine.pairwise_predicate(pred, P, lhs, P, rhs)
        want = np.array([oracle.predicate(pred, allh[l], allh[r]) for l, r in zip(lhs, rhs)])
        assert np.array_equal(got, want), (pred, np.flatnonzero(got != want)[:10])
        seen[pred] = (int(got.sum()), int((~got).sum()))
    assert all(t > 0 and f > 0 for t, f in seen.values()), seen   # both outcomes exercised for every predicate
    assert engine.pairwise_predicate("equals", P, [], P, []).size == 0
    for h in allh:
        oracle.free(h)


def test_pairwise_predicates_vs_reference(engine, ref):
    rng = np.random.default_rng(27)
    hs = [ref.from_sorted(random_bitmap(rng, max_keys=4, key_space=5)) for _ in range(40)]
    hs += [ref.op("and", hs[i], hs[i + 1]) for i in range(20)]
    P = engine.pool_from_serialized([ref.serialize(h) for h in hs])
    n = len(hs)
    lhs = np.concatenate([rng.integers(0, n, 300), np.arange(40, 60), np.arange(10)]).astype(np.uint32)
    rhs = np.concatenate([rng.integers(0, n, 300), np.arange(0, 20), np.arange(10)]).astype(np.uint32)
    for pred in ("intersect", "is_subset", "is_strict_subset", "equals"):
        got = engine.pairwise_predicate(pred, P, lhs, P, rhs)
        want = np.array([ref.predicate(pred, hs[l], hs[r]) for l, r in zip(lhs, rhs)])
        assert np.array_equal(got, want), pred
    for h in hs:
        ref.free(h)


def test_bulk_serialization(engine, oracle):
    """rhip_pool_portable_serialize_many: device-assembled images are byte-identical to the per-bitmap path and to
    the oracle, for every header variant (no runs; runs with < 4 and >= 4 containers; empty) and every payload
    alignment the run-flag bytes can produce."""
    rng = np.random.default_rng(28)
    hs = []
    for nk in list(range(0, 20)) + [33, 64, 65]:          # (n + 7) / 8 odd and even, n < 4 and n >= 4
        for ro in (False, True):
            keys = np.sort(rng.choice(200, nk, replace=False)).astype(np.uint32)
            parts = [(k << np.uint32(16)) | chunk_values(rng, PROFILES[int(rng.integers(0, len(PROFILES)))]).astype(np.uint32)
                     for k in keys]
            v = np.concatenate(parts) if parts else np.zeros(0, np.uint32)
            hs.append(oracle.from_sorted(v, run_optimize=ro))
    bufs = [oracle.serialize(h) for h in hs]
    P = engine.pool_from_serialized(bufs)
    blob, offs = P.serialize_many()
    assert offs[0] == 0 and int(offs[-1]) == blob.size == sum(len(b) for b in bufs)
    raw = blob.tobytes()
    for i, b in enumerate(bufs):
        assert raw[int(offs[i]):int(offs[i + 1])] == b, i
    assert P.serialize_all() == bufs
    # subset, arbitrary order
    ids = rng.permutation(len(bufs))[:17].astype(np.uint32)
    blob2, offs2 = P.serialize_many(ids)
    raw2 = blob2.tobytes()
    for k, i in enumerate(ids):
        assert raw2[int(offs2[k]):int(offs2[k + 1])] == bufs[i]
    # a result pool (slots are upper bounds, directory compacted) through the same path
    lhs = rng.integers(0, len(bufs), 120).astype(np.uint32)
    rhs = rng.integers(0, len(bufs), 120).astype(np.uint32)
    for op in OPS:
        R = engine.pairwise(op, P, lhs, P, rhs)
        allr = R.serialize_all()
        for k in range(0, 120):
            assert allr[k] == R.serialize(k), (op, k)
        oo = oracle.op(op, hs[lhs[5]], hs[rhs[5]])
        assert allr[5] == oracle.serialize(oo)
        oracle.free(oo)
    with pytest.raises(Exception):
        P.serialize_many([1, 1])
    with pytest.raises(Exception):
        P.serialize_many([len(bufs)])
    b0, o0 = P.serialize_many([])
    assert b0.size == 0 and list(o0) == [0]
    for h in hs:
        oracle.free(h)
