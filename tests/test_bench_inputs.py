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


def test_c4_generator_is_valid_portable_and_strided(oracle):
    """rhip_synth_sparse_sizes/_fill: valid portable bitmaps of 32 array containers with 1..512
    values; the (first, stride) form used for sharding b mod G yields exactly the matching bitmaps of the full set."""
    import croaring_amd
    blob, offs = croaring_amd.synth_sparse_portable(0, 1, 48)
    assert offs[0] == 0 and int(offs[-1]) == blob.size
    total = 0
    for b in range(48):
        h = oracle.deserialize(bytes(blob[int(offs[b]):int(offs[b + 1])]))
        assert oracle.validate(h)
        assert oracle.type_counts(h) == (0, 32, 0)
        c = oracle.cardinality(h)
        assert 32 <= c <= 32 * 512
        total += c
        oracle.free(h)
    assert total > 48 * 32 * 100
    for world in (2, 8):
        for rank in (0, world - 1):
            n = (48 - rank + world - 1) // world
            sb, so = croaring_amd.synth_sparse_portable(rank, world, n)
            for i in range(n):
                b = rank + i * world
                assert bytes(sb[int(so[i]):int(so[i + 1])]) == bytes(blob[int(offs[b]):int(offs[b + 1])])


def test_bench_refuses_mislabelled_world():
    """`bench.py --gpus 2` inside a 1-rank environment must fail loudly instead of printing n_gpus: 1."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "refusing" in (p.stderr + p.stdout)
    assert "n_gpus" not in p.stdout
