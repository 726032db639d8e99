"""-m gpu parity gates for the BASELINE configurations that round 1 only checked inside benchmark scripts:
C5 (roaring64, wikileaks-noquotes x 10 buckets), C4 (or_many over the 100 000 seeded sparse bitmaps) against
fixtures produced by the REAL CRoaring (oracle/gen_golden.py c5 c4), and C2 at full size with sampled results
compared byte for byte with the reference library itself (oracle/_ref ships to the GPU box)."""
import os
import zlib

import numpy as np
import pytest

from util import GOLD, OPS, all_pairs, c5_inputs, crc

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ C5
@pytest.fixture(scope="module")
def c5(engine):
    gold = np.load(os.path.join(GOLD, "c5_wikileaks64_pairs.npz"))
    bufs = c5_inputs()
    assert [crc(b) for b in bufs] == list(gold["in_crc"]), "C5 inputs drifted from the fixture"
    pool = engine.pool_from_serialized64(bufs)
    assert pool.is64 and len(pool) == 200
    return pool, gold, bufs


@pytest.mark.parametrize("op", OPS)
def test_c5_roaring64_all_pairs(engine, c5, op):
    """roaring64_bitmap_{and,or,xor,andnot} (roaring64.c:1332-1373, 1541-1593, 1663-1720, 1809-1861) over all 19 900
    pairs: cardinality, portable size and crc32 of every serialized result equal the reference's."""
    pool, gold, _ = c5
    lhs, rhs = all_pairs(200)
    assert np.array_equal(np.stack([lhs, rhs], 1), gold["pairs"].astype(np.uint32))
    res = engine.pairwise(op, pool, lhs, pool, rhs)
    cards = res.cardinalities()
    assert np.array_equal(cards, gold[f"{op}_card"]), f"C5 {op}: cardinalities differ"
    assert np.array_equal(engine.pairwise_cardinality(op, pool, lhs, pool, rhs), cards)
    blob, offs = res.serialize_many()
    sizes = np.diff(offs).astype(np.uint32)
    assert np.array_equal(sizes, gold[f"{op}_size"]), f"C5 {op}: portable sizes differ"
    raw = blob.tobytes()
    bad = [k for k in range(len(lhs)) if zlib.crc32(raw[int(offs[k]):int(offs[k + 1])]) != gold[f"{op}_crc"][k]]
    assert not bad, f"C5 {op}: {len(bad)} of {len(lhs)} results differ from roaring64_bitmap_{op}, first {bad[:5]}"
    # the single-image path agrees with the bulk path
    for k in (0, 777, 19899):
        assert res.serialize(k) == raw[int(offs[k]):int(offs[k + 1])]


def test_c5_roaring64_union_200(engine, oracle, c5):
    """The 200-way union (reference: left fold of roaring64_bitmap_or_inplace, no C many-way API -- SURVEY G9):
    set-equal (L1) to the reference's fold, equal cardinality, valid."""
    pool, gold, _ = c5
    got = engine.or_many(pool)
    assert int(got.cardinalities()[0]) == int(gold["fold_or_card"][0])
    hg = oracle.deserialize64(got.serialize(0))
    hw = oracle.deserialize64(bytes(gold["fold_or"]))
    x = oracle.op64("xor", hg, hw)
    assert oracle.cardinality64(x) == 0
    for h in (hg, hw, x):
        oracle.free64(h)



def test_c5_roaring64_union_sharded_8(engine, oracle, c5):
    """BASELINE configs[4] "8 GPU aggregation" on 8 LOGICAL shards of one device ("64-bit: identical",
    owner = key mod G on the 48-bit container keys): bitmaps b mod 8 -> partial chunks (rhip_many_partials on a
    roaring64 pool) -> every chunk to the owner of its key -> rhip_many_finalize(is64).  The owners' results have
    disjoint keys; their union is set-equal to the reference's 200-way fold (Roaring64Map::fastunion /
    roaring64_bitmap_or_inplace, cpp/roaring/roaring64map.hh:1549-1670, src/roaring64.c:1541-1593), and the
    cardinalities add up to the fixture's."""
    import torch
    from croaring_amd.distributed import _DevArray, shard_ids
    pool, gold, _ = c5
    G = 8
    parts = [engine.many_partials("or", pool, shard_ids(len(pool), s, G)) for s in range(G)]
    engine.synchronize()
    K = torch.cat([torch.as_tensor(_DevArray(p.d_keys, (p.n_keys,)), device="cuda") for p in parts])
    W = torch.cat([torch.as_tensor(_DevArray(p.d_words, (p.n_keys, 1024)), device="cuda") for p in parts])
    assert int(K.max()) >= 1 << 16, "C5 keys must exercise the high-32 part of the 48-bit key"
    card, owned = 0, []
    for owner in range(G):
        sel = (K % G) == owner
        k, w = K[sel].contiguous(), W[sel].contiguous()
        torch.cuda.synchronize()
        res = engine.many_finalize("or", True, k.numel(), k.data_ptr(), w.data_ptr())
        assert res.is64
        vals, _ = res.to_values()
        assert np.all(((vals >> np.uint64(16)) % np.uint64(G)) == np.uint64(owner)), "an owner holds a key it does not own"
        card += int(res.cardinalities()[0])
        owned.append(res.serialize(0))
    assert card == int(gold["fold_or_card"][0])
    whole = engine.or_many(engine.pool_from_serialized64(owned))  # disjoint keys: a pass-through union
    hg = oracle.deserialize64(whole.serialize(0))
    hw = oracle.deserialize64(bytes(gold["fold_or"]))
    x = oracle.op64("xor", hg, hw)
    assert oracle.cardinality64(x) == 0
    for h in (hg, hw, x):
        oracle.free64(h)
    for p in parts:
        p.free()


# ------------------------------------------------------------------ C4
@pytest.fixture(scope="module")
def c4(engine):
    import croaring_amd
    gold = np.load(os.path.join(GOLD, "c4_or_many.npz"))
    n = int(gold["n_bitmaps"][0])
    blob, offs = croaring_amd.synth_sparse_portable(0, 1, n)
    assert blob.size == int(gold["in_bytes"][0]) and zlib.crc32(blob) == int(gold["in_crc"][0]), \
        "C4 generator drifted from the fixture"
    pool = engine.pool_from_blob(blob, offs)
    assert len(pool) == n and pool.type_counts() == (0, 32 * n, 0)
    return pool, gold, n


@pytest.mark.parametrize("n", [1000, 10000, 100000])
def test_c4_or_many_full(engine, c4, n):
    """roaring_bitmap_or_many (roaring.c:775-790) over the first n of the 100 000 seeded sparse bitmaps:
    cardinality, portable size and crc32 of the serialized result equal the reference's."""
    pool, gold, total = c4
    res = engine.or_many(pool, None if n == total else np.arange(n, dtype=np.uint32))
    want_card, want_size, want_crc = (int(x) for x in gold[f"or_many_{n}"])
    assert int(res.cardinalities()[0]) == want_card
    s = res.serialize(0)
    assert len(s) == want_size and zlib.crc32(s) == want_crc, f"C4 or_many over {n}: bytes differ from the reference"


def test_c4_xor_many_and_shards(engine, c4):
    """xor_many cardinality vs the reference, and the SURVEY §8e pipeline on 8 logical shards (b mod 8): partial
    chunks -> key owner -> finalize; the owners' cardinalities sum to the reference's or_many cardinality."""
    import torch
    from croaring_amd.distributed import _DevArray, shard_ids
    pool, gold, total = c4
    ids = np.arange(1000, dtype=np.uint32)
    assert int(engine.xor_many(pool, ids).cardinalities()[0]) == int(gold["xor_many_1000_card"][0])
    G = 8
    parts = [engine.many_partials("or", pool, shard_ids(total, s, G)) for s in range(G)]
    engine.synchronize()
    K = torch.cat([torch.as_tensor(_DevArray(p.d_keys, (p.n_keys,)), device="cuda") for p in parts])
    W = torch.cat([torch.as_tensor(_DevArray(p.d_words, (p.n_keys, 1024)), device="cuda") for p in parts])
    card = 0
    for owner in range(G):
        sel = (K % G) == owner
        k, w = K[sel].contiguous(), W[sel].contiguous()
        torch.cuda.synchronize()
        res = engine.many_finalize("or", False, k.numel(), k.data_ptr(), w.data_ptr())
        card += int(res.cardinalities()[0])
    assert card == int(gold[f"or_many_{total}"][0])
    for p in parts:
        p.free()


# ------------------------------------------------------------------ C2 at full size, bytes vs the reference itself
def test_c2_full_size_sampled_bytes(engine, ref):
    """BASELINE config C2 at FULL size (256 x 4096 bitset containers, 8 GiB): results of the bench schedule, sampled,
    are byte-identical to what CRoaring computes from the same serialized operands ('a sampled subset
    bit-for-bit'); every op, cardinalities included."""
    from bench import SEED, schedule
    pool = engine.pool_synth_bitset(256, 4096, SEED)
    lhs, rhs = schedule(0, 250, 256)
    sample = [3, 77, 131, 249]
    hs = {}
    for b in sorted({int(lhs[k]) for k in sample} | {int(rhs[k]) for k in sample}):
        hs[b] = ref.deserialize(pool.serialize(b))
    for op in OPS:
        res = engine.pairwise(op, pool, lhs, pool, rhs)
        cards = res.cardinalities()
        for k in sample:
            want = ref.op(op, hs[int(lhs[k])], hs[int(rhs[k])])
            assert int(cards[k]) == ref.cardinality(want), (op, k)
            assert res.serialize(k) == ref.serialize(want), f"C2 full size: {op} pair {k} differs from CRoaring"
            ref.free(want)
        del res
    for h in hs.values():
        ref.free(h)


def test_c2_full_size_all_cardinalities(engine, ref):
    """SURVEY 8d for C2: 'the oracle checks a sampled subset bit-for-bit and ALL ops by cardinality'.  The 250 pairs of
    the bench schedule over the full 8 GiB pool, every op: the cardinality of every result bitmap (the materialising
    call) and of the cardinality-only call equals CRoaring's roaring_bitmap_{and,or,xor,andnot}_cardinality on the same
    serialized operands (all 256 bitmaps go through the reference: 8 GiB of host memory)."""
    from bench import SEED, schedule
    pool = engine.pool_synth_bitset(256, 4096, SEED)
    lhs, rhs = schedule(0, 250, 256)
    hs = [ref.deserialize(pool.serialize(b)) for b in range(256)]
    try:
        for op in OPS:
            want = np.array([ref.op_cardinality(op, hs[int(a)], hs[int(b)]) for a, b in zip(lhs, rhs)], dtype=np.uint64)
            res = engine.pairwise(op, pool, lhs, pool, rhs)
            assert np.array_equal(res.cardinalities().astype(np.uint64), want), op
            del res
            assert np.array_equal(engine.pairwise_cardinality(op, pool, lhs, pool, rhs).astype(np.uint64), want), op + "_cardinality"
    finally:
        for h in hs:
            ref.free(h)
