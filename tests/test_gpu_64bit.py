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
