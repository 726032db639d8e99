"""The synthetic workload generators of bench.py produce valid portable bitmaps (checked by the oracle),
so what the GPU is timed on is what the CPU baseline is timed on."""
import numpy as np

import bench
from gen_inputs import splitmix64


def test_c2_bitset_bitmap_is_valid_portable(oracle):
    words = splitmix64((bench.SEED + 3) & (2**64 - 1), 16 * 1024)
    buf = bench.portable_bitset_bitmap(words)
    h = oracle.deserialize(buf)
    assert oracle.validate(h)
    assert oracle.type_counts(h) == (16, 0, 0)
    assert oracle.serialize(h) == buf
    assert oracle.cardinality(h) == int(np.bitwise_count(words).sum())
    oracle.free(h)


def test_c2_schedule_matches_survey():
    lhs, rhs = bench.schedule(0, 1000, 256)
    k = np.arange(1000)
    assert np.array_equal(lhs, k % 256) and np.array_equal(rhs, (k * 97 + 1) % 256)


def test_c4_shard_is_valid_portable(oracle):
    blob, offs, lens = bench.c4_shard(40, 123)
    total = 0
    for o, l in zip(offs, lens):
        h = oracle.deserialize(bytes(blob[int(o):int(o + l)]))
        assert oracle.validate(h)
        b, a, r = oracle.type_counts(h)
        assert (b, a, r) == (0, 32, 0)
        total += oracle.cardinality(h)
        oracle.free(h)
    assert int(offs[-1] + lens[-1]) == blob.size and total > 40 * 32
