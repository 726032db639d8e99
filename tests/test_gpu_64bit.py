"""GPU parity for 64-bit bitmaps (roaring64_bitmap_and/or/xor/andnot, portable 64-bit format)."""
import os

import numpy as np
import pytest

from gen_inputs import random_bitmap
from util import GOLD, OPS

pytestmark = pytest.mark.gpu


def rand64(rng):
    highs = rng.choice(6, int(rng.integers(0, 4)), replace=False)
    parts = [(np.uint64(int(h) * 7 + 1) << np.uint64(32)) | random_bitmap(rng, max_keys=4, key_space=6).astype(np.uint64)
             for h in highs]
    return np.sort(np.concatenate(parts)) if parts else np.zeros(0, np.uint64)


def test_64bit_pairwise_and_many(engine, oracle):
    rng = np.random.default_rng(64)
    hs = [oracle.from_sorted64(rand64(rng)) for _ in range(24)]
    bufs = [oracle.serialize64(h) for h in hs]
    pool = engine.pool_from_serialized64(bufs)
    assert pool.is64
    for i, b in enumerate(bufs):
        assert pool.serialize(i) == b
    lhs = np.arange(24, dtype=np.uint32)
    rhs = (lhs * 7 + 5) % 24
    for op in OPS:
        res = engine.pairwise(op, pool, lhs, pool, rhs)
        cards = res.cardinalities()
        for k in range(24):
            oo = oracle.op64(op, hs[lhs[k]], hs[rhs[k]])
            assert res.serialize(k) == oracle.serialize64(oo), (op, k)
            assert cards[k] == oracle.cardinality64(oo)
            oracle.free64(oo)
    got = engine.or_many(pool).serialize(0)
    want = oracle.or_many64(hs)
    hg = oracle.deserialize64(got)
    assert oracle.cardinality64(hg) == oracle.cardinality64(want)
    # set equality through the 64-bit portable image re-typed by the oracle's own or-fold
    x = oracle.op64("xor", hg, want)
    assert oracle.cardinality64(x) == 0
    for h in hs + [want, hg, x]:
        oracle.free64(h)


def test_64bit_reference_fixtures(engine):
    """tests/testdata/64map*.bin round-trip through the device pool byte-identically."""
    names = ("64map32bitvals.bin", "64mapspreadvals.bin", "64maphighvals.bin", "64mapempty.bin")
    bufs = [open(os.path.join(GOLD, n), "rb").read() for n in names]
    pool = engine.pool_from_serialized64(bufs)
    for i, b in enumerate(bufs):
        assert pool.serialize(i) == b, names[i]


def many64_grouping_body(make_engine, oracle, monkeypatch):
    """The many-way path over 48-bit keys groups by a counting sort over the DENSE ids of the pool's distinct keys (its key
    dictionary, round 6); RHIP_MANY_DICT=0 keeps the radix sort of (key, descriptor) pairs, which also serves pools of
    more than 65 536 distinct keys.  Same bytes either way -- or_many / xor_many, whole pool and a shuffled selection --
    and the sets of the oracle's folds."""
    rng = np.random.default_rng(640)
    hs = [oracle.from_sorted64(rand64(rng)) for _ in range(40)]
    bufs = [oracle.serialize64(h) for h in hs]
    ids = rng.permutation(40).astype(np.uint32)[:29]
    outs = {}
    for dict_on in ("1", "0"):
        monkeypatch.setenv("RHIP_MANY_DICT", dict_on)
        eng = make_engine()
        try:
            pool = eng.pool_from_serialized64(bufs)
            outs[dict_on] = [eng.or_many(pool).serialize(0), eng.xor_many(pool).serialize(0),
                             eng.or_many(pool, ids).serialize(0), eng.xor_many(pool, ids).serialize(0),
                             eng.or_many(pool).serialize(0)]  # (twice: the dictionary is cached with the pool)
        finally:
            eng.close()
    assert outs["1"] == outs["0"]
    want = oracle.or_many64(hs)
    hg = oracle.deserialize64(outs["1"][0])
    x = oracle.op64("xor", hg, want)
    assert oracle.cardinality64(x) == 0
    acc = oracle.from_sorted64(np.zeros(0, np.uint64))
    for i in ids:
        t = oracle.op64("xor", acc, hs[i])
        oracle.free64(acc)
        acc = t
    hx = oracle.deserialize64(outs["1"][3])
    d = oracle.op64("xor", hx, acc)
    assert oracle.cardinality64(d) == 0
    for h in hs + [want, hg, x, acc, hx, d]:
        oracle.free64(h)


def test_64bit_many_grouping_paths(oracle, monkeypatch):
    import croaring_amd
    many64_grouping_body(lambda: croaring_amd.Engine(0), oracle, monkeypatch)
