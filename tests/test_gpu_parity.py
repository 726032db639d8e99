"""GPU parity tests: the HIP engine (through the C ABI) vs the reference's golden results
(tests/golden/*, produced by the real CRoaring) and vs the CPU oracle, bit for bit."""
import os

import numpy as np
import pytest

from util import DATASETS, OPS, all_pairs, crc, load_bundle, load_pairs, synth_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def synth(oracle):
    singles, many = synth_inputs()
    gold = np.load(__import__("os").path.join(__import__("util").GOLD, "synth_mixed.npz"))
    bufs = []
    for a, b in singles:
        for v in (a, b):
            h = oracle.from_sorted(v)
            bufs.append(oracle.serialize(h))
            oracle.free(h)
    assert [crc(b) for b in bufs] == list(gold["in_crc"]), "synthetic inputs drifted from the fixture"
    return bufs, many, gold


@pytest.mark.parametrize("op", OPS)
def test_synth_every_type_pair(engine, oracle, synth, op):
    """All 14x14 container-profile pairs + 120 random multi-key pairs: byte-identical portable
    serialization vs the reference (crc/size/card fixture) and vs the oracle (full bytes)."""
    bufs, _, gold = synth
    pool = engine.pool_from_serialized(bufs)
    n = len(bufs) // 2
    lhs, rhs = np.arange(n, dtype=np.uint32) * 2, np.arange(n, dtype=np.uint32) * 2 + 1
    res = engine.pairwise(op, pool, lhs, pool, rhs)
    cards = res.cardinalities()
    bad = []
    for k in range(n):
        got = res.serialize(k)
        oa, ob = oracle.deserialize(bufs[2 * k]), oracle.deserialize(bufs[2 * k + 1])
        oo = oracle.op(op, oa, ob)
        want = oracle.serialize(oo)
        for h in (oa, ob, oo):
            oracle.free(h)
        if got != want or crc(got) != gold[f"{op}_crc"][k] or len(got) != gold[f"{op}_size"][k] \
                or cards[k] != gold[f"{op}_card"][k]:
            bad.append(k)
    assert not bad, f"{op}: {len(bad)} mismatching pairs, first {bad[:10]}"
    c2 = engine.pairwise_cardinality(op, pool, lhs, pool, rhs)
    assert np.array_equal(c2, gold[f"{op}_card"].astype(np.uint64))


@pytest.mark.parametrize("name", DATASETS)
@pytest.mark.parametrize("op", OPS)
def test_realdata_all_pairs(engine, name, op):
    """Every unordered pair of a realdata set: cardinality, portable size and crc32 of the
    serialized result equal the reference's (SURVEY §8d C1/C3 checksums included)."""
    bufs = load_bundle(name)
    gold = load_pairs(name)
    pool = engine.pool_from_serialized(bufs)
    lhs, rhs = all_pairs(len(bufs))
    assert np.array_equal(np.stack([lhs, rhs], 1), gold["pairs"].astype(np.uint32))
    res = engine.pairwise(op, pool, lhs, pool, rhs)
    cards = res.cardinalities()
    assert int(cards.sum()) == int(gold[f"{op}_card"].astype(np.uint64).sum())
    assert np.array_equal(cards, gold[f"{op}_card"].astype(np.uint64))
    cc = engine.pairwise_cardinality(op, pool, lhs, pool, rhs)
    assert np.array_equal(cc, cards)
    bad = 0
    for k in range(len(lhs)):
        s = res.serialize(k)
        if len(s) != gold[f"{op}_size"][k] or crc(s) != gold[f"{op}_crc"][k]:
            bad += 1
    assert bad == 0, f"{name} {op}: {bad} of {len(lhs)} results differ from the reference"


@pytest.mark.parametrize("name", DATASETS)
def test_realdata_many(engine, oracle, name):
    """or_many / xor_many over a whole dataset: BYTE-identical to the reference's roaring_bitmap_or_many /
    roaring_bitmap_xor_many (L2: the container types of both folds are replayed, full unions and run accumulators
    included)."""
    bufs = load_bundle(name)
    gold = load_pairs(name)
    pool = engine.pool_from_serialized(bufs)
    for nm, fn in (("or_many", engine.or_many), ("xor_many", engine.xor_many)):
        got = fn(pool).serialize(0)
        want = bytes(gold[nm])
        hg, hw = oracle.deserialize(got), oracle.deserialize(want)
        assert oracle.validate(hg)
        assert np.array_equal(oracle.to_array(hg), oracle.to_array(hw)), f"{name} {nm}: set mismatch"
        assert got == want, f"{name} {nm}: container types differ from roaring_bitmap_{nm}"
        oracle.free(hg)
        oracle.free(hw)
    # roaring_bitmap_or_many_heap: the tournament's own container types (on wikileaks-noquotes and census-income they differ
    # from or_many's: SURVEY G11)
    assert engine.or_many_heap(pool).serialize(0) == bytes(gold["or_many_heap"]), f"{name}: or_many_heap bytes"


def test_synth_many(engine, oracle, synth):
    _, many, gold = synth
    for k, vs in enumerate(many):
        hs = [oracle.from_sorted(v) for v in vs]
        bufs = [oracle.serialize(h) for h in hs]
        pool = engine.pool_from_serialized(bufs)
        for nm, fn in (("or_many", engine.or_many), ("xor_many", engine.xor_many)):
            got = fn(pool).serialize(0)
            hg = oracle.deserialize(got)
            ow = (oracle.or_many if nm == "or_many" else oracle.xor_many)(hs)
            assert oracle.validate(hg)
            assert np.array_equal(oracle.to_array(hg), oracle.to_array(ow)), f"group {k} {nm}"
            assert oracle.cardinality(hg) == gold[f"{nm}_card"][k]
            assert got == oracle.serialize(ow), f"group {k}: {nm} bytes differ from the oracle"
            assert crc(got) == gold[f"{nm}_crc"][k], f"group {k}: {nm} bytes differ from CRoaring"
            oracle.free(hg)
            oracle.free(ow)
        got = engine.or_many_heap(pool).serialize(0)
        oh = oracle.or_many_heap(hs)
        assert got == oracle.serialize(oh), f"group {k}: or_many_heap bytes differ from the oracle"
        assert crc(got) == gold["or_many_heap_crc"][k], f"group {k}: or_many_heap bytes differ from CRoaring"
        oracle.free(oh)
        for h in hs:
            oracle.free(h)


def test_pairwise_multi(engine, oracle, synth):
    """rhip_pairwise_multi: several ops over one pair list planned as ONE batch; bitmap o * npairs + k of the result
    is byte-identical to rhip_pairwise(ops[o]) -- every class kernel reads the op from its work items."""
    bufs, _, _ = synth
    pool = engine.pool_from_serialized(bufs)
    n = len(bufs)
    rng = np.random.default_rng(99)
    lhs = rng.integers(0, n, 500).astype(np.uint32)
    rhs = rng.integers(0, n, 500).astype(np.uint32)
    single = {op: engine.pairwise(op, pool, lhs, pool, rhs) for op in OPS}
    for ops in (list(OPS), ["or", "and"], ["andnot", "xor", "andnot"], ["xor"]):
        res = engine.pairwise_multi(ops, pool, lhs, pool, rhs)
        assert len(res) == len(ops) * lhs.size
        blob, offs = res.serialize_many()
        for o, op in enumerate(ops):
            wb, wo = single[op].serialize_many()
            lo, hi = int(offs[o * lhs.size]), int(offs[(o + 1) * lhs.size])
            assert hi - lo == wb.size and np.array_equal(blob[lo:hi], wb), (ops, op)
            assert np.array_equal(offs[o * lhs.size:(o + 1) * lhs.size + 1] - offs[o * lhs.size], wo), (ops, op)
    # the result is an ordinary pool: chained on the device
    res = engine.pairwise_multi(["and", "or"], pool, lhs, pool, rhs)
    k = np.arange(lhs.size, dtype=np.uint32)
    back = engine.pairwise("and", res, k + lhs.size, res, k)  # (a | b) & (a & b) == a & b
    assert np.array_equal(back.serialize_many()[0], single["and"].serialize_many()[0])


def test_class_stats(engine, oracle, synth):
    """rhip_last_class_stats: the per-kernel split adds up to the batch totals of rhip_last_stats (SURVEY §8d
    algorithmic bytes), for every op."""
    bufs, _, _ = synth
    pool = engine.pool_from_serialized(bufs)
    n = len(bufs)
    rng = np.random.default_rng(77)
    lhs = rng.integers(0, n, 400).astype(np.uint32)
    rhs = rng.integers(0, n, 400).astype(np.uint32)
    engine.set_class_stats(True)
    try:
        for op in OPS:
            engine.pairwise(op, pool, lhs, pool, rhs)
            st, cs = engine.last_stats(), engine.last_class_stats()
            assert sum(v["bytes_in"] for v in cs.values()) == st["bytes_in"], (op, cs, st)
            assert sum(v["bytes_out"] for v in cs.values()) == st["bytes_out"], (op, cs, st)
            assert sum(v["items"] for k, v in cs.items() if k != "k_copy") == st["matched_pairs"], (op, cs, st)
            assert cs["k_copy"]["items"] == st["passthrough"]
    finally:
        engine.set_class_stats(False)


def sparse_many_body(eng, oracle, n=600, worlds=(1, 3)):
    """The C4 generator at test size (32 array containers of 1..512 values per bitmap, key 0 in every
    bitmap): or_many byte-identical to the oracle, xor_many set-equal, and the DENSE sharded pipeline on `world`
    logical shards -- rhip_many_partials_dense per shard, the all-to-all done by hand, rhip_many_finalize_dense per
    owner -- reproduces the union.  Device tables are torch tensors on the engine's device."""
    import torch
    import croaring_amd
    from croaring_amd.distributed import dense_block, shard_ids
    blob, offs = croaring_amd.synth_sparse_portable(0, 1, n)
    bufs = [bytes(blob[int(offs[i]):int(offs[i + 1])]) for i in range(n)]
    hs = [oracle.deserialize(b) for b in bufs]
    pool = eng.pool_from_blob(blob, offs)
    want_or, want_xor = oracle.or_many(hs), oracle.xor_many(hs)
    got = eng.or_many(pool).serialize(0)
    assert got == oracle.serialize(want_or), "sparse or_many: bytes differ from the oracle"
    assert eng.last_stats()["bytes_in"] == pool.payload_bytes()
    assert eng.xor_many(pool).serialize(0) == oracle.serialize(want_xor), "sparse xor_many: bytes differ from the oracle"
    sub = np.arange(5, n, 7, dtype=np.uint32)  # a selection: same through ids
    ws = oracle.or_many([hs[i] for i in sub])
    assert eng.or_many(pool, sub).serialize(0) == oracle.serialize(ws)
    oracle.free(ws)
    dev = eng.torch_device()
    for op, want in (("or", want_or), ("xor", want_xor)):
        wv = oracle.to_array(want)
        for world in worlds:
            B = dense_block(4096, world)
            with eng.torch_stream():
                sends = []
                for r in range(world):
                    t = torch.empty((world * B, 1024), dtype=torch.int64, device=dev)
                    eng.many_partials_dense(op, pool, shard_ids(n, r, world) if world > 1 else None, 4096, world, t.data_ptr())
                    sends.append(t)
                vals = []
                for owner in range(world):  # the all-to-all: owner receives block `owner` of every source's table
                    recv = torch.cat([sends[s][owner * B:(owner + 1) * B] for s in range(world)]).contiguous()
                    res = eng.many_finalize_dense(op, False, world, owner, B, recv.data_ptr())
                    h = oracle.deserialize(res.serialize(0))
                    assert oracle.validate(h)
                    v = oracle.to_array(h)
                    assert np.all((v >> 16) % world == owner)
                    vals.append(v)
                    oracle.free(h)
            assert np.array_equal(np.sort(np.concatenate(vals)), wv), (op, world)
    # a key beyond the table is refused by the owner's finalize
    with eng.torch_stream():
        t = torch.empty((16, 1024), dtype=torch.int64, device=dev)
        eng.many_partials_dense("or", pool, None, 16, 1, t.data_ptr())
        with pytest.raises(Exception):
            eng.many_finalize_dense("or", False, 1, 0, 16, t.data_ptr())
        # an ABANDONED pipeline (stage 1 failed on the device, its finalize never ran) is reported by synchronize and
        # does not fail the next, healthy pipeline: the error word is tagged with the pipeline that raised it
        eng.many_partials_dense("or", pool, None, 16, 1, t.data_ptr())
        with pytest.raises(Exception):
            eng.synchronize()
        B = dense_block(4096, 1)
        t2 = torch.empty((B, 1024), dtype=torch.int64, device=dev)
        eng.many_partials_dense("or", pool, None, 4096, 1, t2.data_ptr())  # (joins the open pipeline: still failed)
        with pytest.raises(Exception):
            eng.many_finalize_dense("or", False, 1, 0, B, t2.data_ptr())
        eng.many_partials_dense("or", pool, None, 4096, 1, t2.data_ptr())  # a fresh pipeline
        res = eng.many_finalize_dense("or", False, 1, 0, B, t2.data_ptr())
        assert res.serialize(0) == oracle.serialize(want_or)
        eng.synchronize()
    for h in hs + [want_or, want_xor]:
        oracle.free(h)


@pytest.mark.gpu
def test_sparse_many_and_dense_shards(engine, oracle):
    sparse_many_body(engine, oracle, n=3000, worlds=(1, 2, 8))


def test_bitset_only_synthetic_pool(engine, oracle):
    """SURVEY §8d C2 at test size: device-generated splitmix64 pool == host restatement, and
    pairwise results on it are byte-identical to the oracle."""
    from gen_inputs import splitmix64
    nb, nc, seed = 6, 40, 0x9E3779B97F4A7C15
    pool = engine.pool_synth_bitset(nb, nc, seed)
    assert pool.type_counts() == (nb * nc, 0, 0)
    hs = []
    for b in range(nb):
        words = splitmix64((seed + b) & (2**64 - 1), nc * 1024)
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")
        vals = np.flatnonzero(bits).astype(np.uint32)
        h = oracle.from_sorted(vals, run_optimize=False)
        assert oracle.serialize(h) == pool.serialize(b)
        hs.append(h)
    lhs = np.array([0, 1, 2, 3, 4, 5, 0], np.uint32)
    rhs = np.array([1, 2, 3, 4, 5, 0, 0], np.uint32)
    for op in OPS:
        res = engine.pairwise(op, pool, lhs, pool, rhs)
        st = engine.last_stats()
        assert st["n_bitset_pairs"] == 7 * nc
        for k in range(len(lhs)):
            oo = oracle.op(op, hs[lhs[k]], hs[rhs[k]])
            assert res.serialize(k) == oracle.serialize(oo), (op, k)
            oracle.free(oo)
    for h in hs:
        oracle.free(h)


def test_edge_cases(engine, oracle):
    """Empty bitmaps, empty batches, self-pairs, disjoint keys."""
    e = oracle.from_sorted(np.zeros(0, np.uint32))
    a = oracle.from_sorted(np.arange(0, 200000, 3, dtype=np.uint32))
    b = oracle.from_sorted(np.arange(1 << 20, (1 << 20) + 5000, dtype=np.uint32))
    hs = [e, a, b]
    bufs = [oracle.serialize(h) for h in hs]
    pool = engine.pool_from_serialized(bufs)
    assert len(engine.pairwise("and", pool, [], pool, [])) == 0
    idx = [(i, j) for i in range(3) for j in range(3)]
    lhs = np.array([i for i, _ in idx], np.uint32)
    rhs = np.array([j for _, j in idx], np.uint32)
    for op in OPS:
        res = engine.pairwise(op, pool, lhs, pool, rhs)
        for k, (i, j) in enumerate(idx):
            oo = oracle.op(op, hs[i], hs[j])
            assert res.serialize(k) == oracle.serialize(oo), (op, i, j)
            oracle.free(oo)
    assert engine.or_many(pool, []).serialize(0) == bufs[0]
    assert engine.or_many(pool, [1]).serialize(0) == bufs[1]
    # chained: (a | b) & a == a   -- results are pools and feed the next op without leaving HBM
    u = engine.pairwise("or", pool, [1], pool, [2])
    back = engine.pairwise("and", u, [0], pool, [1])
    assert back.serialize(0) == bufs[1]
    for h in hs:
        oracle.free(h)
    with pytest.raises(Exception):
        engine.pool_from_serialized([b"\x00\x01\x02\x03garbage"])
    with pytest.raises(Exception):
        engine.pairwise("and", pool, [7], pool, [0])


@pytest.mark.parametrize("op", ["or", "xor"])
@pytest.mark.parametrize("shards", [2, 4, 8])
def test_sharded_many_logical_shards(engine, oracle, op, shards):
    """SURVEY §8e on ONE device: G logical shards -> per-shard partial chunks (rhip_many_partials) ->
    chunks routed to key owners -> owner combine + canonicalise (rhip_many_finalize).  The union of the
    owners' results must equal or_many / xor_many over the whole set."""
    import torch
    from croaring_amd.distributed import _DevArray, shard_ids
    bufs = load_bundle("census-income")[:64] + load_bundle("wikileaks-noquotes")[:32]
    hs = [oracle.deserialize(b) for b in bufs]
    pool = engine.pool_from_serialized(bufs)
    parts = [engine.many_partials(op, pool, shard_ids(len(bufs), s, shards)) for s in range(shards)]
    engine.synchronize()
    keys = [torch.as_tensor(_DevArray(p.d_keys, (p.n_keys,)), device="cuda") for p in parts if p.n_keys]
    words = [torch.as_tensor(_DevArray(p.d_words, (p.n_keys, 1024)), device="cuda") for p in parts if p.n_keys]
    K, W = torch.cat(keys), torch.cat(words)
    got_vals = []
    for owner in range(shards):
        sel = (K % shards) == owner
        k, w = K[sel].contiguous(), W[sel].contiguous()
        torch.cuda.synchronize()
        res = engine.many_finalize(op, False, k.numel(), k.data_ptr() if k.numel() else 0,
                                   w.data_ptr() if k.numel() else 0)
        h = oracle.deserialize(res.serialize(0))
        assert oracle.validate(h)
        v = oracle.to_array(h)
        assert np.all((v >> 16) % shards == owner)
        got_vals.append(v)
        oracle.free(h)
    want = (oracle.or_many if op == "or" else oracle.xor_many)(hs)
    assert np.array_equal(np.sort(np.concatenate(got_vals)), oracle.to_array(want))
    single = (engine.or_many if op == "or" else engine.xor_many)(pool)
    hsingle = oracle.deserialize(single.serialize(0))
    assert np.array_equal(oracle.to_array(hsingle), oracle.to_array(want))
    for h in hs + [want, hsingle]:
        oracle.free(h)
    for p in parts:
        p.free()


def test_or_many_full_container_typing(engine, oracle):
    """The run-vs-bitset choice for FULL unions depends on the fold order in the reference
    (roaring.c:2529-2548 vs 2619-2647); every ordering of a small adversarial set must match."""
    import itertools
    full = np.arange(65536, dtype=np.uint32)
    half_a = np.arange(0, 65536, 2, dtype=np.uint32)
    half_b = np.arange(1, 65536, 2, dtype=np.uint32)
    lo = np.arange(0, 40000, dtype=np.uint32)
    hi = np.arange(30000, 65536, dtype=np.uint32)
    few = np.array([5, 77, 4000], dtype=np.uint32)
    members = {
        "fullrun": oracle.from_sorted(full),                        # run {0,65535}
        "fullbits": oracle.from_sorted(full, run_optimize=False),   # bitset with card 65536
        "evens": oracle.from_sorted(half_a, run_optimize=False),    # bitset
        "odds": oracle.from_sorted(half_b, run_optimize=False),     # bitset
        "lo": oracle.from_sorted(lo),                               # run
        "hi": oracle.from_sorted(hi),                               # run
        "few": oracle.from_sorted(few),                             # array
    }
    names = list(members)
    bufs = [oracle.serialize(members[n]) for n in names]
    pool = engine.pool_from_serialized(bufs)
    checked = 0
    for r in (2, 3, 4):
        for combo in itertools.permutations(range(len(names)), r):
            if r == 4 and checked % 5:  # thin out the 840 4-permutations
                checked += 1
                continue
            checked += 1
            want = oracle.or_many([members[names[i]] for i in combo])
            got = engine.or_many(pool, list(combo)).serialize(0)
            assert got == oracle.serialize(want), [names[i] for i in combo]
            oracle.free(want)
    for h in members.values():
        oracle.free(h)


def test_many_sharded_nccl_world1(engine, oracle):
    """croaring_amd.distributed.many_sharded end to end on the RCCL backend with a 1-rank group
    (the only world size one GPU allows): partials -> exchange -> finalize -> gather."""
    import torch
    import torch.distributed as dist
    from croaring_amd.distributed import gather_serialized, many_sharded
    os_env = __import__("os").environ
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os_env["MASTER_ADDR"] = "127.0.0.1"
    os_env["MASTER_PORT"] = str(port)
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        bufs = load_bundle("census1881")[:50]
        hs = [oracle.deserialize(b) for b in bufs]
        pool = engine.pool_from_serialized(bufs)
        for op, fn in (("or", oracle.or_many), ("xor", oracle.xor_many)):
            owned = many_sharded(engine, pool, op)
            blob = gather_serialized(engine, owned)
            hg, want = oracle.deserialize(blob), fn(hs)
            assert oracle.validate(hg)
            assert np.array_equal(oracle.to_array(hg), oracle.to_array(want)), op
            oracle.free(hg)
            oracle.free(want)
        for h in hs:
            oracle.free(h)
    finally:
        if created:
            dist.destroy_process_group()


def test_full_size_properties(engine):
    """BASELINE config C2 at FULL size (pool of 256 bitmaps x 4096 bitset containers = 8 GiB): properties
    that need no oracle -- idempotence, self-inverse, inclusion-exclusion, commutativity checksums."""
    from bench import SEED, schedule
    pool = engine.pool_synth_bitset(256, 4096, SEED)
    cards = pool.cardinalities()
    assert pool.type_counts() == (256 * 4096, 0, 0)
    assert abs(float(cards.mean()) / (4096 * 65536) - 0.5) < 1e-3       # density 0.5
    ids = np.arange(0, 256, 8, dtype=np.uint32)                          # 32 bitmaps
    # a & a == a, a | a == a, a ^ a == {}, a \ a == {}
    for op, want in (("and", cards[ids]), ("or", cards[ids]), ("xor", 0 * cards[ids]), ("andnot", 0 * cards[ids])):
        r = engine.pairwise(op, pool, ids, pool, ids)
        assert np.array_equal(r.cardinalities(), want), op
        if op in ("xor", "andnot"):
            assert r.n_containers == 0                                   # empties are dropped
        else:
            assert r.serialize(3) == pool.serialize(int(ids[3]))         # byte-identical to the operand
    lhs, rhs = schedule(0, 32, 256)
    c_and = engine.pairwise("and", pool, lhs, pool, rhs).cardinalities()
    c_or = engine.pairwise("or", pool, lhs, pool, rhs).cardinalities()
    c_xor = engine.pairwise("xor", pool, lhs, pool, rhs).cardinalities()
    c_andnot = engine.pairwise("andnot", pool, lhs, pool, rhs).cardinalities()
    assert np.array_equal(c_and + c_or, cards[lhs] + cards[rhs])         # |A n B| + |A u B| = |A| + |B|
    assert np.array_equal(c_xor, c_or - c_and)
    assert np.array_equal(c_andnot, cards[lhs] - c_and)
    for op, c in (("and", c_and), ("or", c_or), ("xor", c_xor), ("andnot", c_andnot)):
        assert np.array_equal(engine.pairwise_cardinality(op, pool, lhs, pool, rhs), c), op
    assert np.array_equal(engine.pairwise("and", pool, rhs, pool, lhs).cardinalities(), c_and)  # commutes
    # or_many over 16 half-dense bitmaps: every chunk fills up (65536 * (1 - 2^-16) expected)
    u = engine.or_many(pool, np.arange(16, dtype=np.uint32))
    assert u.n_containers == 4096 and int(u.cardinalities()[0]) > 4096 * 65530


def test_directory_and_payload_extremes(engine, oracle):
    """Key-merge and slot-sizing extremes: a bitmap with all 65536 container keys, a non-efficient run
    container whose payload exceeds 8192 bytes (passes through unchanged, roaring.c:914-941), and
    operands whose directories differ by four orders of magnitude."""
    import struct
    # (1) every key present, one value each -> 65536 array containers
    allkeys = (np.arange(65536, dtype=np.uint32) << 16) | (np.arange(65536, dtype=np.uint32) % 7)
    a = oracle.from_sorted(allkeys)
    # (2) a run container with 3000 runs (12000-byte payload): valid, just not run-efficient -- built by hand
    starts = np.arange(3000, dtype=np.uint32) * 20
    runs = np.stack([starts, np.full(3000, 9, np.uint32)], 1).astype(np.uint16)      # [s, s+9]
    payload = struct.pack("<H", 3000) + runs.tobytes()
    card = 3000 * 10
    big = struct.pack("<I", 12347 | (0 << 16)) + b"\x01" + struct.pack("<HH", 3, card - 1) + payload
    hb = oracle.deserialize(big)
    assert oracle.validate(hb) and oracle.serialize(hb) == big
    # (3) a small bitmap
    c = oracle.from_sorted(np.array([5, (3 << 16) | 7, (3 << 16) | 100, (70 << 16) | 1], dtype=np.uint32))
    hs = [a, hb, c]
    bufs = [oracle.serialize(h) for h in hs]
    pool = engine.pool_from_serialized(bufs)
    assert pool.serialize(1) == big
    idx = [(i, j) for i in range(3) for j in range(3)]
    lhs = np.array([i for i, _ in idx], np.uint32)
    rhs = np.array([j for _, j in idx], np.uint32)
    for op in OPS:
        res = engine.pairwise(op, pool, lhs, pool, rhs)
        cards = engine.pairwise_cardinality(op, pool, lhs, pool, rhs)
        for k, (i, j) in enumerate(idx):
            oo = oracle.op(op, hs[i], hs[j])
            assert res.serialize(k) == oracle.serialize(oo), (op, i, j)
            assert cards[k] == oracle.cardinality(oo)
            oracle.free(oo)
    for nm, fn, of in (("or", engine.or_many, oracle.or_many), ("xor", engine.xor_many, oracle.xor_many)):
        want = of(hs)
        assert fn(pool).serialize(0) == oracle.serialize(want), nm
        oracle.free(want)
    for h in hs:
        oracle.free(h)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_randomized_pools(engine, oracle, seed):
    """Randomised differential test through the C ABI: 120 multi-key bitmaps of mixed container profiles,
    600 random pairs x 4 ops (+ cardinalities) against the oracle, byte level; random or_many subsets."""
    from gen_inputs import random_bitmap
    rng = np.random.default_rng(seed)
    hs = [oracle.from_sorted(random_bitmap(rng, max_keys=10, key_space=14)) for _ in range(120)]
    bufs = [oracle.serialize(h) for h in hs]
    pool = engine.pool_from_serialized(bufs)
    lhs = rng.integers(0, 120, 600).astype(np.uint32)
    rhs = rng.integers(0, 120, 600).astype(np.uint32)
    for op in OPS:
        res = engine.pairwise(op, pool, lhs, pool, rhs)
        cards = engine.pairwise_cardinality(op, pool, lhs, pool, rhs)
        bad = []
        for k in range(600):
            oo = oracle.op(op, hs[lhs[k]], hs[rhs[k]])
            if res.serialize(k) != oracle.serialize(oo) or cards[k] != oracle.cardinality(oo):
                bad.append(k)
            oracle.free(oo)
        assert not bad, f"seed {seed} {op}: {len(bad)} mismatches, first {bad[:5]}"
    for _ in range(12):
        ids = rng.choice(120, int(rng.integers(2, 40)), replace=False).astype(np.uint32)
        want = oracle.or_many([hs[i] for i in ids])
        assert engine.or_many(pool, ids).serialize(0) == oracle.serialize(want)
        oracle.free(want)
        wx = oracle.xor_many([hs[i] for i in ids])
        assert engine.xor_many(pool, ids).serialize(0) == oracle.serialize(wx), "xor_many bytes"
        oracle.free(wx)
    for h in hs:
        oracle.free(h)


def test_array_array_union_boundaries(engine, oracle):
    """array x array or / xor around every size boundary of the typing rule (ca + cb <= 4096 -> array, else by
    cardinality) and of the kernels' 64-lane / 512-value steps: identical / disjoint / interleaved / nested operands."""
    rng = np.random.default_rng(77)
    cases = []
    ident = np.sort(rng.choice(65536, 1020, replace=False))
    cases += [(ident, ident), (ident, ident[::2]), (ident[::2], ident), (ident[:1], ident), (ident, ident[-1:])]
    cases += [(np.arange(0, 2000, 2), np.arange(1, 2001, 2)), (np.arange(0, 1000), np.arange(1000, 2000)),
              (np.arange(1000, 2000), np.arange(0, 1000)), (np.arange(0, 1020), np.arange(0, 1020) + 1),
              (np.arange(0, 4000, 2), np.arange(1, 4001, 2)), (np.arange(0, 2044), np.arange(0, 2044) + 1)]
    for tot in (2, 3, 63, 64, 65, 127, 128, 129, 1000, 2039, 2040, 2041, 2047, 2048, 4088, 4095, 4096, 4097, 5000, 8000):
        for frac in (0.02, 0.5, 0.98):
            ca = max(1, min(4096, int(tot * frac)))
            cb = max(1, min(4096, tot - ca))
            univ = rng.choice(65536, min(65536, int((ca + cb) * rng.choice([1.0, 1.3, 4.0]))), replace=False)
            a = np.sort(rng.choice(univ, min(ca, univ.size), replace=False))
            b = np.sort(rng.choice(univ, min(cb, univ.size), replace=False))
            cases.append((a, b))
    # the rank-merge kernel (short array into a long one): short side 1 / 64 / 65 / 128 / 129 / 192 / 193 / 255 / 256
    # values (four rounds of 64; 255 is its limit: mode 3 packs ~128 new values behind one rank byte counter), sums at
    # 4096 / 4097, shared values (xor deletes them), short values below / above / between all long values, both ends
    for ny, nx in ((1, 1), (1, 4095), (1, 4096), (64, 900), (65, 900), (127, 1000), (128, 1000), (129, 1000), (128, 3968),
                   (128, 3969), (100, 3996), (100, 3997), (2, 7), (128, 128), (128, 129), (40, 4000), (192, 700),
                   (193, 700), (254, 1000), (255, 1000), (256, 1000), (257, 1000), (255, 255), (255, 3841), (255, 3842),
                   (200, 200)):
        big = np.sort(rng.choice(np.arange(300, 65000), nx, replace=False))
        for mode in range(4):
            if mode == 0:
                small = np.sort(rng.choice(65536, ny, replace=False))                     # mostly new values
            elif mode == 1:
                small = np.sort(rng.choice(big, min(ny, nx), replace=False))              # all shared
            elif mode == 2:
                small = np.unique(np.concatenate([rng.choice(big, min(ny // 2 + 1, nx), replace=False),
                                                  rng.choice(65536, ny, replace=False)]))[:ny]   # mixed
            else:
                small = np.unique(np.concatenate([np.arange(0, ny // 2 + 1), 65535 - np.arange(0, ny // 2 + 1)]))[:ny]
            cases.append((np.sort(small), big))
    cases.append((np.arange(255), np.arange(300, 1300)))        # 255 new values at rank 0: the byte counter's limit
    cases.append((np.arange(65281, 65536), np.arange(300, 1300)))  # ... and at rank nx
    hs = []
    for a, b in cases:
        hs.append(oracle.from_sorted(np.asarray(a, np.uint32) + (7 << 16), run_optimize=False))
        hs.append(oracle.from_sorted(np.asarray(b, np.uint32) + (7 << 16), run_optimize=False))
    bufs = [oracle.serialize(h) for h in hs]
    pool = engine.pool_from_serialized(bufs)
    assert pool.type_counts()[1] == len(hs)          # all array containers
    n = len(cases)
    lhs, rhs = np.arange(n, dtype=np.uint32) * 2, np.arange(n, dtype=np.uint32) * 2 + 1
    for op in ("or", "xor"):
        for l, r in ((lhs, rhs), (rhs, lhs)):
            res = engine.pairwise(op, pool, l, pool, r)
            cards = engine.pairwise_cardinality(op, pool, l, pool, r)
            for k in range(n):
                oo = oracle.op(op, hs[l[k]], hs[r[k]])
                assert res.serialize(k) == oracle.serialize(oo), (op, k, len(cases[k][0]), len(cases[k][1]))
                assert cards[k] == oracle.cardinality(oo)
                oracle.free(oo)
    for h in hs:
        oracle.free(h)


def test_array_filter_probe_boundaries(engine, oracle):
    """and / andnot / and_cardinality with an array operand around every boundary of the two filter kernels: streamed
    side of 1 .. 257 values (k_probe holds at most 256: one to four per lane), probed arrays whose pivot step changes
    (64 / 65 / 128 / 129 / 4096 values), a bitset partner, values at both ends of the u16 range, hits at pivots and
    between them, identical and disjoint operands; both operand orders."""
    rng = np.random.default_rng(99)
    small = [1, 2, 31, 63, 64, 65, 100, 127, 128, 129, 192, 193, 200, 255, 256, 257]
    large = [1, 5, 63, 64, 65, 127, 128, 129, 640, 1000, 4095, 4096]
    cases = []
    for ny in small:
        for nx in large:
            univ = np.sort(rng.choice(65536, min(65536, max(ny, nx) * int(rng.choice([1, 2, 8])) + 8), replace=False))
            x = np.sort(rng.choice(univ, nx, replace=False))
            y = np.sort(rng.choice(univ, ny, replace=False))
            if rng.random() < 0.3:
                y = np.unique(np.concatenate([y[: max(1, ny // 2)], x[:: max(1, nx // max(1, ny // 2))][: ny // 2 + 1]]))[:ny]
            if rng.random() < 0.25:
                x = np.unique(np.concatenate([x, [0, 65535]]))[-nx:] if nx > 2 else x
                y = np.unique(np.concatenate([y, [0, 65535]]))
            cases.append((y, x))
        dense = np.flatnonzero(rng.random(65536) < 0.3)
        cases.append((np.sort(rng.choice(65536, ny, replace=False)), dense))           # array x bitset
        cases.append((np.sort(rng.choice(dense, ny, replace=False)), dense))           # all hits
    ident = np.sort(rng.choice(65536, 128, replace=False))
    cases += [(ident, ident), (ident[:64], ident), (ident, ident[64:]), (np.arange(100), np.arange(100, 1000))]
    hs = []
    for a, b in cases:
        hs.append(oracle.from_sorted(np.asarray(a, np.uint32) + (3 << 16), run_optimize=False))
        hs.append(oracle.from_sorted(np.asarray(b, np.uint32) + (3 << 16), run_optimize=False))
    bufs = [oracle.serialize(h) for h in hs]
    pool = engine.pool_from_serialized(bufs)
    n = len(cases)
    lhs, rhs = np.arange(n, dtype=np.uint32) * 2, np.arange(n, dtype=np.uint32) * 2 + 1
    for op in ("and", "andnot"):
        for l, r in ((lhs, rhs), (rhs, lhs)):
            res = engine.pairwise(op, pool, l, pool, r)
            cards = engine.pairwise_cardinality(op, pool, l, pool, r)
            for k in range(n):
                oo = oracle.op(op, hs[l[k]], hs[r[k]])
                assert res.serialize(k) == oracle.serialize(oo), (op, k, len(cases[k][0]), len(cases[k][1]))
                assert cards[k] == oracle.cardinality(oo), (op, k)
                oracle.free(oo)
    for h in hs:
        oracle.free(h)


def test_tiny_interval_pairs(engine, oracle):
    """Interval pairs in all three size classes of k_ivl (four pairs per wave up to 31 and up to 127 intervals a side,
    one pair per wave beyond): run containers of 1 .. 128 runs and arrays of 1 .. 128 values around the classification
    boundaries (31 / 32 and 127 / 128 intervals per operand; value totals around 512, 1024 and 4096), touching and
    nested intervals, both ends of the u16 range, full containers; four ops + cardinalities, both operand orders."""
    rng = np.random.default_rng(314)

    def runs(k, maxlen):
        cuts = np.sort(rng.choice(65536, 2 * k, replace=False))
        parts = [np.arange(cuts[2 * i], min(cuts[2 * i + 1], cuts[2 * i] + maxlen) + 1) for i in range(k)]
        return np.unique(np.concatenate(parts))

    shapes = []
    for k in (1, 2, 3, 8, 16, 31, 32, 33, 63, 64, 65, 70, 127, 128):
        shapes.append(("run", runs(k, 400)))
        shapes.append(("run", runs(k, 2)))
        shapes.append(("arr", np.sort(rng.choice(65536, k, replace=False))))
        shapes.append(("arr", np.sort(rng.choice(2000, k, replace=False)) + 30000))
    shapes += [("run", np.arange(65536)), ("run", np.arange(0, 33)), ("run", np.arange(0, 32)),
               ("run", np.concatenate([np.arange(0, 10), np.arange(65500, 65536)])),
               ("arr", np.array([0, 65535])), ("arr", np.array([9, 10, 65499, 65500])),
               ("run", np.concatenate([np.arange(100, 200), np.arange(201, 300), np.arange(301, 5000)])),
               # 256 + 256 = 512 values (four per wave), 256 + 257 = 513 (one per wave); 481 / 482 + a 31-value array
               ("run", np.arange(1000, 1256)), ("run", np.arange(1100, 1356)), ("run", np.arange(5000, 5257)),
               ("run", np.concatenate([np.arange(40, 300), np.arange(1200, 1421)])),
               ("run", np.concatenate([np.arange(40, 300), np.arange(1200, 1422)])),
               ("arr", np.arange(31) * 41 + 50),
               # 512 + 512 = 1024 and 512 + 513 values; 2048 + 2048 = 4096 and 2048 + 2049
               ("run", np.arange(2000, 2512)), ("run", np.arange(2300, 2812)), ("run", np.arange(9000, 9513)),
               ("run", np.arange(20000, 22048)), ("run", np.arange(21000, 23048)), ("run", np.arange(40000, 42049))]
    hs = [oracle.from_sorted(np.asarray(v, np.uint32) + (9 << 16), run_optimize=(kind == "run")) for kind, v in shapes]
    bufs = [oracle.serialize(h) for h in hs]
    pool = engine.pool_from_serialized(bufs)
    n = len(hs)
    lhs, rhs = np.meshgrid(np.arange(n, dtype=np.uint32), np.arange(n, dtype=np.uint32))
    lhs, rhs = lhs.ravel().copy(), rhs.ravel().copy()
    for op in OPS:
        res = engine.pairwise(op, pool, lhs, pool, rhs)
        cards = engine.pairwise_cardinality(op, pool, lhs, pool, rhs)
        blob, offs = res.serialize_many()
        raw = blob.tobytes()
        bad = []
        for k in range(lhs.size):
            oo = oracle.op(op, hs[lhs[k]], hs[rhs[k]])
            if raw[int(offs[k]):int(offs[k + 1])] != oracle.serialize(oo) or cards[k] != oracle.cardinality(oo):
                bad.append((int(lhs[k]), int(rhs[k])))
            oracle.free(oo)
        assert not bad, f"{op}: {len(bad)} mismatching pairs, first {bad[:6]}"
    for h in hs:
        oracle.free(h)


def test_prepared_pair_lists(engine, oracle, synth):
    """rhip_pairlist_*: a pair list prepared once (validated, summed, resident on the device); batches over it are
    byte-identical to the ad-hoc calls -- single op, several ops in one batch, cardinalities, two batches in flight,
    the all-pairs / successive generators (benchmarks/benchmark.cpp:2035-2091) -- and the list re-validates itself when
    an operand pool changes under it (in-place update)."""
    bufs, _, _ = synth
    pool = engine.pool_from_serialized(bufs)
    n = len(bufs)
    rng = np.random.default_rng(4242)
    lhs = rng.integers(0, n, 300).astype(np.uint32)
    rhs = rng.integers(0, n, 300).astype(np.uint32)
    pl = engine.pairlist(pool, lhs, pool, rhs)
    assert len(pl) == 300 and all(np.array_equal(a, b) for a, b in zip(pl.pairs(), (lhs, rhs)))
    for op in OPS:
        want = engine.pairwise(op, pool, lhs, pool, rhs)
        got = engine.pairwise_list(op, pl)
        assert np.array_equal(got.serialize_many()[0], want.serialize_many()[0]), op
        assert np.array_equal(engine.pairwise_list_cardinality(op, pl), want.cardinalities()), op
        got = engine.pairwise_list(op, pl, reuse=got)  # recycled result pool
        assert np.array_equal(got.serialize_many()[0], want.serialize_many()[0]), op
    wm = engine.pairwise_multi(["xor", "and", "or"], pool, lhs, pool, rhs)
    gm = engine.pairwise_list(["xor", "and", "or"], pl)
    assert np.array_equal(gm.serialize_many()[0], wm.serialize_many()[0])
    # the plan is kept with the list (round 6): a repeated (ops, form) starts at its class kernels, same bytes; more
    # distinct plans than the list keeps (six) evict the least recently used; two batches in flight share one plan
    cached = os.environ.get("RHIP_PLAN_CACHE", "1") != "0"
    wants = {op: engine.pairwise(op, pool, lhs, pool, rhs).serialize_many()[0] for op in OPS}
    for rnd in range(2):
        for op in OPS:
            got = engine.pairwise_list(op, pl)
            if rnd == 1:  # (round 0: some of the four were evicted by the nine plans made above)
                assert engine.plan_cached() == cached, op
            assert np.array_equal(got.serialize_many()[0], wants[op]), (rnd, op)
    gm2 = engine.pairwise_list(["xor", "and", "or"], pl)
    assert np.array_equal(gm2.serialize_many()[0], wm.serialize_many()[0])
    for op in OPS:
        c1 = engine.pairwise_list_cardinality(op, pl)
        c2 = engine.pairwise_list_cardinality(op, pl)
        assert np.array_equal(c1, c2) and np.array_equal(c1, engine.pairwise_cardinality(op, pool, lhs, pool, rhs)), op
    ba, bb, bc = (engine.pairwise_list_begin("xor", pl) for _ in range(3))
    for b in (bb, bc, ba):
        assert np.array_equal(b.end().serialize_many()[0], wants["xor"])
    b1 = engine.pairwise_list_begin("or", pl)
    b2 = engine.pairwise_list_begin(["andnot"], pl)
    pl_small = engine.pairlist(pool, lhs[:5], pool, rhs[:5])
    b3 = engine.pairwise_list_begin("and", pl_small)
    pl_small.free()  # deferred: b3 still reads it
    r2, r1, r3 = b2.end(), b1.end(), b3.end()
    assert np.array_equal(r1.serialize_many()[0], engine.pairwise("or", pool, lhs, pool, rhs).serialize_many()[0])
    assert np.array_equal(r2.serialize_many()[0], engine.pairwise("andnot", pool, lhs, pool, rhs).serialize_many()[0])
    assert np.array_equal(r3.serialize_many()[0], engine.pairwise("and", pool, lhs[:5], pool, rhs[:5]).serialize_many()[0])
    # generators
    ns = 48
    small = engine.pool_from_serialized(bufs[100:100 + ns])
    ap = engine.pairlist_all_pairs(small)
    L, R = all_pairs(ns)
    assert all(np.array_equal(a, b) for a, b in zip(ap.pairs(), (L, R)))
    assert np.array_equal(engine.pairwise_list("xor", ap).serialize_many()[0],
                          engine.pairwise("xor", small, L, small, R).serialize_many()[0])
    sc = engine.pairlist_successive(pool)
    k = np.arange(n - 1, dtype=np.uint32)
    assert all(np.array_equal(a, b) for a, b in zip(sc.pairs(), (k, k + 1)))
    assert np.array_equal(engine.pairwise_list("andnot", sc).serialize_many()[0],
                          engine.pairwise("andnot", pool, k, pool, k + 1).serialize_many()[0])
    # an operand updated in place under the list: the next batch over it sees the new pool (bounds re-taken)
    p2 = engine.pool_from_serialized(bufs)
    pl2 = engine.pairlist(p2, lhs, p2, rhs)
    engine.pairwise_list("or", pl2)
    engine.pairwise_list("or", pl2)
    assert engine.plan_cached() == cached
    upd = np.arange(0, n, 3, dtype=np.uint32)
    engine.pairwise_inplace("or", p2, upd, p2, (upd + 1) % n)
    got = engine.pairwise_list("or", pl2)
    assert not engine.plan_cached(), "a plan cached before an in-place update of an operand pool must not be used after it"
    want = engine.pairwise("or", p2, lhs, p2, rhs)
    assert np.array_equal(got.serialize_many()[0], want.serialize_many()[0])
    got = engine.pairwise_list("or", pl2)
    assert engine.plan_cached() == cached and np.array_equal(got.serialize_many()[0], want.serialize_many()[0])
    hs = [oracle.deserialize(b) for b in bufs]
    for i in upd[:20]:
        o = oracle.op("or", hs[i], hs[(i + 1) % n])
        assert p2.serialize(int(i)) == oracle.serialize(o)
        oracle.free(o)
    for h in hs:
        oracle.free(h)
    # errors: an index out of range is refused when the list is made
    with pytest.raises(Exception):
        engine.pairlist(pool, np.array([0, n], np.uint32), pool, np.array([0, 0], np.uint32))
    assert len(engine.pairwise_list("and", engine.pairlist(pool, np.zeros(0, np.uint32), pool, np.zeros(0, np.uint32)))) == 0
    # a list that has been freed while a batch still read it cannot start another batch; a NULL list is refused
    pl3 = engine.pairlist(pool, lhs[:4], pool, rhs[:4])
    bb = engine.pairwise_list_begin("or", pl3)
    h3 = pl3.h
    pl3.free()                         # deferred
    pl3.h = h3
    with pytest.raises(Exception):
        engine.pairwise_list("and", pl3)
    pl3.h = None
    assert len(bb.end()) == 4          # (the last batch over it releases the list)
    import ctypes
    assert not engine.lib.rhip_pairwise_list(engine.h, 1, (ctypes.c_int * 1)(0), None, None)
    # a pool freed while a list holds it as an operand stays alive until the list goes (the list pins its pools); the
    # list of another context is refused
    p4 = engine.pool_from_serialized(bufs[:40])
    l4, r4 = np.arange(39, dtype=np.uint32), np.arange(1, 40, dtype=np.uint32)
    want4 = engine.pairwise("xor", p4, l4, p4, r4).serialize_many()[0]
    pl4 = engine.pairlist(p4, l4, p4, r4)
    h4 = p4.h
    engine.lib.rhip_pool_free(h4)      # deferred: pl4 pins the pool
    p4.h = None
    assert np.array_equal(engine.pairwise_list("xor", pl4).serialize_many()[0], want4)
    pl4.free()                         # the list goes, and with it the pool


def test_tiny_passthrough_containers(engine, oracle):
    """Pass-through containers of sparse bitmaps: the pool averages well under 96 payload bytes per container, so k_copy
    takes SIXTEEN items per wave (four lanes each).  Bitmaps with disjoint key sets (everything passes through under or /
    xor / andnot), 1 .. 32-value arrays and 1 .. 16-run containers mostly, with a few 60 .. 120-value arrays (the
    quarter-wave rounds) and one bitset per ten bitmaps (the whole-wave round) among them; queue lengths around the
    multiples of 16; chained once (the result pool is not tiny: four per wave)."""
    rng = np.random.default_rng(2718)
    vals = []
    for b in range(30):
        keys = np.sort(rng.choice(400, int(rng.integers(1, 50)), replace=False))
        parts = []
        for k in keys:
            kind = rng.integers(0, 20)
            if kind == 0 and b % 10 == 3:
                v = np.sort(rng.choice(65536, 5000, replace=False))
            elif kind <= 2:
                v = np.sort(rng.choice(65536, int(rng.integers(60, 121)), replace=False))
            elif kind <= 6:
                st = np.sort(rng.choice(60000, int(rng.integers(1, 17)), replace=False))
                v = np.unique(np.concatenate([np.arange(x, x + int(rng.integers(2, 40))) for x in st]))
            else:
                v = np.sort(rng.choice(65536, int(rng.integers(1, 33)), replace=False))
            parts.append(v.astype(np.uint32) + (np.uint32(k) << 16))
        vals.append(np.concatenate(parts))
    hs = [oracle.from_sorted(v, run_optimize=True) for v in vals]
    pool = engine.pool_from_serialized([oracle.serialize(h) for h in hs])
    assert pool.payload_bytes() / max(1, sum(pool.type_counts())) <= 96
    n = len(hs)
    lhs, rhs = np.meshgrid(np.arange(n, dtype=np.uint32), np.arange(n, dtype=np.uint32))
    lhs, rhs = lhs.ravel().copy(), rhs.ravel().copy()
    for op in OPS:
        for cut in (lhs.size, 17, 16, 1):
            res = engine.pairwise(op, pool, lhs[:cut], pool, rhs[:cut])
            blob, offs = res.serialize_many()
            raw = blob.tobytes()
            bad = []
            for k in range(cut):
                oo = oracle.op(op, hs[lhs[k]], hs[rhs[k]])
                if raw[int(offs[k]):int(offs[k + 1])] != oracle.serialize(oo):
                    bad.append((int(lhs[k]), int(rhs[k])))
                oracle.free(oo)
            assert not bad, f"{op} ({cut} pairs): {len(bad)} mismatching pairs, first {bad[:6]}"
        idx = np.arange(min(len(res), 8), dtype=np.uint32)  # chained: (a op b) | a
        res2 = engine.pairwise("or", res, idx, pool, lhs[idx])
        for k in idx:
            o1 = oracle.op(op, hs[lhs[k]], hs[rhs[k]])
            o2 = oracle.op("or", o1, hs[lhs[k]])
            assert res2.serialize(int(k)) == oracle.serialize(o2), (op, int(k))
            oracle.free(o1); oracle.free(o2)
    for h in hs:
        oracle.free(h)


def test_pairwise_placed(engine, oracle, synth):
    """Engine.pairwise_placed: the result pool kept after the start-up tries holds the batch's results and can be
    recycled; the other tries are gone."""
    bufs = synth[0]
    hs = [oracle.deserialize(b) for b in bufs]
    pool = engine.pool_from_serialized(bufs)
    n = len(hs)
    lhs = np.arange(n, dtype=np.uint32)
    rhs = np.roll(lhs, 3)
    (res, spare), ms = engine.pairwise_placed("or", pool, lhs, pool, rhs, tries=3, keep=2)
    assert len(ms) == 3 and all(m >= 0 for m in ms)
    spare = engine.pairwise("and", pool, lhs, pool, rhs, reuse=spare)
    assert len(spare) == n
    res = engine.pairwise("xor", pool, lhs, pool, rhs, reuse=res)
    for k in range(n):
        oo = oracle.op("xor", hs[lhs[k]], hs[rhs[k]])
        assert res.serialize(k) == oracle.serialize(oo), k
        oracle.free(oo)
    for h in hs:
        oracle.free(h)


def test_long_interval_lists(engine, oracle):
    """k_genw's long-list interval path (a run operand, lists too long for the k_ivl classes, at most 2 032 intervals
    together) and its neighbours: run containers of 100 .. 1 500 runs against arrays of 300 .. 3 000 values and against
    each other, sums around the 2 032 limit (2 031 / 2 032 / 2 033: the last one takes the image path), dense operands
    whose or / xor must come out as a bitset (the interval path hands those over to the image path), full and
    near-full runs, both ends of the u16 range; four ops + cardinalities, both operand orders."""
    rng = np.random.default_rng(2718)

    def runs(k, maxlen):
        cuts = np.sort(rng.choice(65536, 2 * k, replace=False))
        parts = [np.arange(cuts[2 * i], min(cuts[2 * i + 1], cuts[2 * i] + maxlen) + 1) for i in range(k)]
        return np.unique(np.concatenate(parts))

    def exact_runs(k):  # exactly k runs of two values, gaps of at least one
        starts = np.sort(rng.choice(65536 // 4, k, replace=False)).astype(np.int64) * 4
        return np.concatenate([starts, starts + 1])

    shapes = [("run", runs(100, 30)), ("run", runs(100, 400)), ("run", runs(300, 8)), ("run", runs(600, 4)),
              ("run", runs(1000, 2)), ("run", runs(1500, 2)),
              ("arr", np.sort(rng.choice(65536, 300, replace=False))), ("arr", np.sort(rng.choice(65536, 874, replace=False))),
              ("arr", np.sort(rng.choice(65536, 1500, replace=False))), ("arr", np.sort(rng.choice(65536, 3000, replace=False))),
              ("arr", np.sort(rng.choice(3000, 1900, replace=False)) + 20000),
              ("arr", np.concatenate([[0, 1, 2], np.sort(rng.choice(60000, 1000, replace=False)) + 100, [65534, 65535]])),
              ("run", np.sort(exact_runs(1016))), ("arr", np.sort(rng.choice(65536, 1015, replace=False))),
              ("arr", np.sort(rng.choice(65536, 1016, replace=False))), ("arr", np.sort(rng.choice(65536, 1017, replace=False))),
              ("run", np.concatenate([np.arange(0, 30000), np.arange(30002, 65536)])), ("run", np.arange(65536)),
              ("run", np.concatenate([np.arange(40 * i, 40 * i + 25) for i in range(1600)])[:40000]),
              ("run", runs(400, 60))]
    hs = [oracle.from_sorted(np.asarray(np.sort(v), np.uint32) + (5 << 16), run_optimize=(kind == "run")) for kind, v in shapes]
    bufs = [oracle.serialize(h) for h in hs]
    pool = engine.pool_from_serialized(bufs)
    n = len(hs)
    lhs, rhs = np.meshgrid(np.arange(n, dtype=np.uint32), np.arange(n, dtype=np.uint32))
    lhs, rhs = lhs.ravel().copy(), rhs.ravel().copy()
    for op in OPS:
        res = engine.pairwise(op, pool, lhs, pool, rhs)
        cards = engine.pairwise_cardinality(op, pool, lhs, pool, rhs)
        blob, offs = res.serialize_many()
        raw = blob.tobytes()
        bad = []
        for k in range(lhs.size):
            oo = oracle.op(op, hs[lhs[k]], hs[rhs[k]])
            if raw[int(offs[k]):int(offs[k + 1])] != oracle.serialize(oo) or cards[k] != oracle.cardinality(oo):
                bad.append((int(lhs[k]), int(rhs[k])))
            oracle.free(oo)
        assert not bad, f"{op}: {len(bad)} mismatching pairs, first {bad[:6]}"
    for h in hs:
        oracle.free(h)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1", "2", "fork", "nomerge"])
def test_explicit_unit_arrays(oracle, synth, monkeypatch, mode):
    """Paths the size of a batch selects, forced on the same small inputs (big batches reach them by themselves).
    Small bitmaps plan on implicit units, four per wave up to 64 containers a bitmap: RHIP_EXPLICIT_UNITS=1 forces the
    staged unit arrays of the general path, =2 implicit units with one unit per wave.  Small batches keep every class
    kernel in ONE launch (k_classes): RHIP_FORK_MIN_MB=0 forks the stand-alone kernels onto the auxiliary streams,
    RHIP_MERGE_CLASSES=0 launches them one by one on the main stream."""
    import croaring_amd
    if mode == "fork":
        monkeypatch.setenv("RHIP_FORK_MIN_MB", "0")
    elif mode == "nomerge":
        monkeypatch.setenv("RHIP_MERGE_CLASSES", "0")
    else:
        monkeypatch.setenv("RHIP_EXPLICIT_UNITS", mode)
    eng = croaring_amd.Engine()
    try:
        test_edge_cases(eng, oracle)
        for op in OPS:
            test_synth_every_type_pair(eng, oracle, synth, op)
        if mode == "nomerge":
            test_tiny_passthrough_containers(eng, oracle)  # k_copy itself at sixteen items per wave
        if mode == "1":
            test_prepared_pair_lists(eng, oracle, synth)   # a prepared list whose batches stage explicit units
    finally:
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("vmm", ["1", "0"])
def test_arena_placement_forced(oracle, synth, monkeypatch, vmm):
    """The measured placement of result arenas forced onto SMALL arenas with RHIP_ARENA_PLACE_MIN_MB=0 -- by address
    (place_arena_va, round 6: one hipMemCreate'd allocation mapped at the positions of an address window, probed with
    k_place_probe against the operand pool at each, left at the fastest) and by candidates (RHIP_ARENA_VMM=0, round 4:
    several allocations probed, the fastest kept): results byte-identical, recycled pools keep working, the probe rates
    are reported.  (At full size the C2 tests and bench.py go through it by themselves.)"""
    import croaring_amd
    monkeypatch.setenv("RHIP_ARENA_PLACE_MIN_MB", "0")
    monkeypatch.setenv("RHIP_ARENA_TRIES", "4")
    monkeypatch.setenv("RHIP_ARENA_VMM", vmm)
    monkeypatch.setenv("RHIP_ARENA_VA_WINDOW_MB", "12")  # six positions 2 MiB apart: every other one, then the best one's neighbours
    monkeypatch.setenv("RHIP_ARENA_VA_STEP_MB", "2")
    eng = croaring_amd.Engine()
    try:
        arena_placement_body(eng, oracle, synth)
    finally:
        eng.close()


def arena_placement_body(eng, oracle, synth):
    bufs, _, _ = synth
    pool = eng.pool_from_serialized(bufs)
    n = len(bufs)
    rng = np.random.default_rng(31)
    lhs = rng.integers(0, n, 200).astype(np.uint32)
    rhs = rng.integers(0, n, 200).astype(np.uint32)
    res = None
    for op in OPS:
        res = eng.pairwise(op, pool, lhs, pool, rhs, reuse=res)  # (the recycled arena grows under or / xor: placed again)
        blob, offs = res.serialize_many()
        raw = blob.tobytes()
        for k in range(0, 200, 7):
            oa, ob = oracle.deserialize(bufs[lhs[k]]), oracle.deserialize(bufs[rhs[k]])
            oo = oracle.op(op, oa, ob)
            assert raw[int(offs[k]):int(offs[k + 1])] == oracle.serialize(oo), (op, k)
            for h in (oa, ob, oo):
                oracle.free(h)
    rates = eng.last_placement()
    assert 1 <= len(rates) <= 96 and all(r > 0 for r in rates), rates  # (single chunks, compositions, then RHIP_ARENA_TRIES candidates -- as many again while all are slow -- and address positions)
    b1 = eng.pairwise_begin("xor", pool, lhs, pool, rhs)  # a fresh result pool while another batch is in flight
    b2 = eng.pairwise_begin("or", pool, lhs, pool, rhs)
    r2, r1 = b2.end(), b1.end()
    assert np.array_equal(r1.serialize_many()[0], eng.pairwise("xor", pool, lhs, pool, rhs).serialize_many()[0])
    assert np.array_equal(r2.serialize_many()[0], eng.pairwise("or", pool, lhs, pool, rhs).serialize_many()[0])


def grouped_body(eng, oracle, synth):
    """Everything with an image-class item, through the X-grouped queues (k_filter_g / k_union_g)."""
    test_edge_cases(eng, oracle)
    for op in OPS:
        test_synth_every_type_pair(eng, oracle, synth, op)
    test_pairwise_multi(eng, oracle, synth)
    test_array_filter_probe_boundaries(eng, oracle)
    test_array_array_union_boundaries(eng, oracle)
    test_class_stats(eng, oracle, synth)
    test_batches_in_flight(eng, oracle, synth)
    # both operands from DIFFERENT pools (the X index space then has an A part and a B part)
    bufs, _, _ = synth
    n = len(bufs)
    pa, pb = eng.pool_from_serialized(bufs[: n // 2 + 3]), eng.pool_from_serialized(bufs[n // 3:])
    hs = [oracle.deserialize(b) for b in bufs]
    rng = np.random.default_rng(77)
    lhs = rng.integers(0, len(pa), 400).astype(np.uint32)
    rhs = rng.integers(0, len(pb), 400).astype(np.uint32)
    for op in OPS:
        res = eng.pairwise(op, pa, lhs, pb, rhs)
        cards = eng.pairwise_cardinality(op, pa, lhs, pb, rhs)
        for k in range(len(lhs)):
            want = oracle.op(op, hs[lhs[k]], hs[n // 3 + rhs[k]])
            assert res.serialize(k) == oracle.serialize(want), (op, k)
            assert cards[k] == oracle.cardinality(want), (op, k)
            oracle.free(want)
    for h in hs:
        oracle.free(h)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["group", "groupfork"])
def test_grouped_queues(oracle, synth, monkeypatch, mode):
    """RHIP_GROUP_X=2 forces the X-grouped image queues (round 4; picked by themselves for all-pairs batches, where a
    container meets many partners) on the small inputs, unforked and forked."""
    import croaring_amd
    monkeypatch.setenv("RHIP_GROUP_X", "2")
    if mode == "groupfork":
        monkeypatch.setenv("RHIP_FORK_MIN_MB", "0")
    eng = croaring_amd.Engine()
    try:
        grouped_body(eng, oracle, synth)
    finally:
        eng.close()


def usmall_variants_body(eng, oracle, synth):
    for op in ("or", "xor"):
        test_synth_every_type_pair(eng, oracle, synth, op)
    test_array_array_union_boundaries(eng, oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("gp8", ["0", "1"])
def test_usmall_lds_variants(oracle, synth, monkeypatch, gp8):
    """k_usmall is built twice (deleted-before counts as 16-bit words: five workgroups per CU; as bytes: six) and the host
    picks per batch; RHIP_USMALL_GP8 pins either, RHIP_MERGE_CLASSES=0 keeps the small test batches out of the merged
    launch (which always runs the 16-bit body)."""
    import croaring_amd
    monkeypatch.setenv("RHIP_USMALL_GP8", gp8)
    monkeypatch.setenv("RHIP_MERGE_CLASSES", "0")
    eng = croaring_amd.Engine()
    try:
        usmall_variants_body(eng, oracle, synth)
    finally:
        eng.close()


@pytest.mark.gpu
def test_batches_in_flight(engine, oracle, synth):
    """rhip_pairwise_begin / _end: four batches of different ops and pair lists in flight at once, ended out of order,
    give byte-for-byte what the synchronous call gives; a fifth begin, a pending result as operand or as `reuse`, and a
    double end are refused; the engine keeps working afterwards."""
    from croaring_amd import RoaringHipError
    bufs, _, _ = synth
    pool = engine.pool_from_serialized(bufs)
    n = len(bufs)
    rng = np.random.default_rng(2024)
    jobs = []
    for op in OPS:
        lhs = rng.integers(0, n, 300 + 50 * len(jobs)).astype(np.uint32)
        rhs = rng.integers(0, n, lhs.size).astype(np.uint32)
        jobs.append((op, lhs, rhs))
    want = [engine.pairwise(op, pool, l, pool, r).serialize_many() for op, l, r in jobs]
    batches = [engine.pairwise_begin(op, pool, l, pool, r) for op, l, r in jobs]
    with pytest.raises(RoaringHipError):          # RHIP_MAX_BATCHES_IN_FLIGHT = 4
        engine.pairwise_begin("and", pool, jobs[0][1], pool, jobs[0][2])
    got = {}
    for k in (2, 0, 3, 1):                        # any order
        got[k] = batches[k].end()
    with pytest.raises(RoaringHipError):
        batches[0].end()
    for k in range(4):
        blob, offs = got[k].serialize_many()
        assert np.array_equal(offs, want[k][1]) and np.array_equal(blob, want[k][0]), jobs[k][0]
    # an operand of a batch in flight cannot be recycled as somebody's result buffer
    opnd = engine.pairwise("and", pool, jobs[0][1], pool, jobs[0][2])
    ids = np.arange(min(16, len(opnd)), dtype=np.uint32)
    hold = engine.pairwise_begin("or", opnd, ids, opnd, ids)
    with pytest.raises(RoaringHipError):
        engine.pairwise_begin("and", pool, jobs[0][1], pool, jobs[0][2], reuse=opnd)
    assert opnd.h is not None and len(opnd) == jobs[0][1].size, "a refused `reuse` pool must stay the caller's"
    with pytest.raises(RoaringHipError):          # nor can it be updated in place
        engine.pairwise_inplace("or", opnd, ids, opnd, ids)
    # synchronous calls do not need a batch slot: cardinalities with four batches in flight
    more = [engine.pairwise_begin(op, pool, l, pool, r) for op, l, r in jobs[:3]]
    cards = engine.pairwise_cardinality("and", pool, jobs[0][1], pool, jobs[0][2])
    assert cards.size == jobs[0][1].size
    # freeing an operand of a batch in flight is deferred to the end of that batch
    opnd_cards = opnd.cardinalities()[:ids.size].copy()
    opnd.free()
    got_hold = hold.end()
    assert np.array_equal(got_hold.cardinalities(), opnd_cards)
    for k, bt in enumerate(more):
        blob, offs = bt.end().serialize_many()
        assert np.array_equal(blob, want[k][0]) and np.array_equal(offs, want[k][1])
    assert np.array_equal(cards, engine.pairwise("and", pool, jobs[0][1], pool, jobs[0][2]).cardinalities())
    # matched container pairs are a property of the pair list, not of the op (or / xor also plan B-side tiles)
    mp = {}
    for op in OPS:
        engine.pairwise(op, pool, jobs[0][1], pool, jobs[0][2])
        mp[op] = engine.last_stats()["matched_pairs"]
    engine.pairwise_cardinality("and", pool, jobs[0][1], pool, jobs[0][2])
    mp["card"] = engine.last_stats()["matched_pairs"]
    assert len(set(mp.values())) == 1 and mp["and"] > 0, mp
    # a result that is still in flight cannot be recycled by another batch; after its end it can
    spare = engine.pairwise("and", pool, jobs[0][1], pool, jobs[0][2])
    b = engine.pairwise_begin("or", pool, jobs[1][1], pool, jobs[1][2], reuse=spare)
    r1 = b.end()
    blob, offs = r1.serialize_many()
    assert np.array_equal(blob, want[1][0]) and np.array_equal(offs, want[1][1])
    # software pipeline: begin(i+1) before end(i), results recycled two batches later
    res = [None, None]
    prev = None
    for it in range(6):
        op, l, r = jobs[it % 4]
        cur = engine.pairwise_begin(op, pool, l, pool, r, reuse=res[it & 1])
        if prev is not None:
            k, pb = prev
            res[(it - 1) & 1] = pb.end()
            blob, offs = res[(it - 1) & 1].serialize_many()
            assert np.array_equal(blob, want[k][0]) and np.array_equal(offs, want[k][1]), (it, k)
        prev = (it % 4, cur)
    k, pb = prev
    last = pb.end()
    blob, offs = last.serialize_many()
    assert np.array_equal(blob, want[k][0]) and np.array_equal(offs, want[k][1])


def _long_run_bitmap(key: int, n_runs: int) -> bytes:
    """A portable image with ONE run container of n_runs runs {20 k, length 1} -- valid (sorted, non-adjacent), never
    produced by run_optimize once 4 n_runs + 2 exceeds a bitset: the only container whose payload is larger than 8 KiB."""
    import struct
    starts = 20 * np.arange(n_runs, dtype=np.uint32)
    runs = np.empty(2 * n_runs, dtype=np.uint16)
    runs[0::2] = starts.astype(np.uint16)
    runs[1::2] = 1
    card = 2 * n_runs
    return (struct.pack("<I", 12347 | (0 << 16)) + bytes([1]) + struct.pack("<HH", key, card - 1) +
            struct.pack("<H", n_runs) + runs.tobytes())


def test_many_long_run_passthrough(engine, oracle):
    """or_many / xor_many where a single-member group is a run container LONGER than a bitset (3 000 runs = 12 002
    bytes): it passes through unchanged (roaring.c:2660-2676), so its result slot exceeds 8 192 bytes -- the arena bound
    must not assume a bitset per group.  Also as a member of a two-member group (result typed by cardinality)."""
    n_runs = 3000
    img = _long_run_bitmap(7, n_runs)
    h = oracle.deserialize(img)
    assert oracle.validate(h) and oracle.cardinality(h) == 2 * n_runs
    rng = np.random.default_rng(99)
    others = [np.sort(rng.choice(1 << 22, 3000, replace=False)).astype(np.uint32) + np.uint32(16 << 16) for _ in range(5)]
    hs = [oracle.from_sorted(v) for v in others]
    bufs = [oracle.serialize(x) for x in hs] + [img]
    pool = engine.pool_from_serialized(bufs)
    allh = hs + [h]
    got = engine.or_many(pool).serialize(0)
    assert got == oracle.serialize(oracle.or_many(allh)), "or_many with a pass-through run container of 3 000 runs"
    assert engine.xor_many(pool).serialize(0) == oracle.serialize(oracle.xor_many(allh))
    # the long run container meeting a partner with the same key
    mate = oracle.from_sorted((np.arange(0, 60000, 7, dtype=np.uint32) + np.uint32(7 << 16)))
    pool2 = engine.pool_from_serialized(bufs + [oracle.serialize(mate)])
    assert engine.or_many(pool2).serialize(0) == oracle.serialize(oracle.or_many(allh + [mate]))
    assert engine.xor_many(pool2).serialize(0) == oracle.serialize(oracle.xor_many(allh + [mate]))
    # ONE bitmap selected: roaring_bitmap_copy (roaring.c:799-801) -- no repair pass, the inefficient run list stays
    one = np.array([len(bufs) - 1], np.uint32)
    assert engine.xor_many(pool, one).serialize(0) == img and engine.or_many(pool, one).serialize(0) == img
    for x in allh + [mate]:
        oracle.free(x)


def xor_many_typing_body(eng, oracle, iters=60, big=True):
    """roaring_bitmap_xor_many is a fixed left fold (roaring.c:795-809), so its container TYPES are reproducible and
    the engine reproduces them (many_xor_replay): few keys with many members, run-heavy mixes -- keys whose members
    are ALL runs (the result stays a run), runs meeting small / large arrays and bitsets, members that cancel (the
    accumulator is removed and re-cloned), inputs with and without run compression, a selection through ids in a
    shuffled order, and (big) a group of more members than the replay sorts at once."""
    from gen_inputs import random_bitmap, chunk_values
    rng = np.random.default_rng(606)
    mixes = (("runs",), ("runs", "shortruns", "tiny", "single", "edge"), ("runs", "tiny"),
             ("runs", "shortruns", "dense", "sparse"), ("full", "nearfull", "runs", "blocks", "verydense"), None)
    n_run_results = 0
    for it in range(iters):
        profs = mixes[it % len(mixes)]
        n = int(rng.integers(2, 14))
        kw = dict(max_keys=3, key_space=3) if profs is None else dict(max_keys=3, key_space=3, profiles=profs)
        vs = [random_bitmap(rng, **kw) for _ in range(n)]
        if it % 5 == 0 and n > 3:
            vs[2] = vs[0]
            if it % 10 == 0:
                vs[1] = vs[0]
        hs = [oracle.from_sorted(v) for v in vs]
        if it % 3 == 1:
            for h in hs[::2]:
                oracle.remove_run_compression(h)
        pool = eng.pool_from_serialized([oracle.serialize(h) for h in hs])
        want = oracle.xor_many(hs)
        got = eng.xor_many(pool).serialize(0)
        assert got == oracle.serialize(want), f"xor_many typing, case {it} ({profs})"
        n_run_results += int(oracle.type_counts(want)[2])
        ids = rng.permutation(n).astype(np.uint32)[: max(1, n - 1)]
        ws = oracle.xor_many([hs[i] for i in ids])
        assert eng.xor_many(pool, ids).serialize(0) == oracle.serialize(ws), f"xor_many typing through ids, case {it}"
        for h in hs + [want, ws]:
            oracle.free(h)
    assert n_run_results > 0, "no case left a run container: the test lost its point"
    if big:  # 2 600 members under one key (the replay sorts 1 024 tags at a time), every third a run container
        vs = []
        for b in range(2600):
            prof = "runs" if b % 3 == 0 else ("tiny" if b % 3 == 1 else "sparse")
            vs.append(chunk_values(rng, prof).astype(np.uint32) | np.uint32(5 << 16))
        vs += [np.array([1, 2, 3], np.uint32)] * 3  # key 0: a group without runs beside it
        hs = [oracle.from_sorted(v) for v in vs]
        pool = eng.pool_from_serialized([oracle.serialize(h) for h in hs])
        want = oracle.xor_many(hs)
        assert eng.xor_many(pool).serialize(0) == oracle.serialize(want), "xor_many typing, 2 600-member group"
        ids = rng.permutation(len(hs)).astype(np.uint32)
        ws = oracle.xor_many([hs[i] for i in ids])
        assert eng.xor_many(pool, ids).serialize(0) == oracle.serialize(ws), "xor_many typing, 2 600 members shuffled"
        for h in hs + [want, ws]:
            oracle.free(h)


def test_xor_many_fold_typing(engine, oracle):
    xor_many_typing_body(engine, oracle)


def or_many_heap_body(eng, oracle, iters=48):
    """rhip_or_many_heap = roaring_bitmap_or_many_heap (src/roaring_priority_queue.c:200-247), bytes: run-heavy mixes on few
    keys, equal-sized inputs (ties are broken by heap position), empty bitmaps, inputs without run compression, a selection
    through ids, one and no bitmap."""
    from gen_inputs import random_bitmap
    rng = np.random.default_rng(707)
    mixes = (("runs", "shortruns", "tiny", "single", "edge"), ("runs", "tiny"), ("runs", "shortruns", "dense", "sparse"),
             ("sparse", "tiny", "mid", "boundary4096"), ("full", "nearfull", "runs", "blocks", "verydense"), None)
    for it in range(iters):
        profs = mixes[it % len(mixes)]
        n = int(rng.integers(2, 22))
        kw = dict(max_keys=4, key_space=5) if profs is None else dict(max_keys=4, key_space=5, profiles=profs)
        vs = [random_bitmap(rng, **kw) for _ in range(n)]
        if it % 7 == 0:
            vs[int(rng.integers(0, n))] = np.zeros(0, np.uint32)
        if it % 5 == 0 and n > 3:
            vs[2] = vs[0]; vs[3] = vs[0]
        hs = [oracle.from_sorted(v) for v in vs]
        if it % 3 == 1:
            for h in hs[::2]:
                oracle.remove_run_compression(h)
        pool = eng.pool_from_serialized([oracle.serialize(h) for h in hs])
        want = oracle.or_many_heap(hs)
        assert eng.or_many_heap(pool).serialize(0) == oracle.serialize(want), f"or_many_heap, case {it} ({profs})"
        ids = rng.permutation(n).astype(np.uint32)[: max(2, n - 2)]
        ws = oracle.or_many_heap([hs[i] for i in ids])
        assert eng.or_many_heap(pool, ids).serialize(0) == oracle.serialize(ws), f"or_many_heap through ids, case {it}"
        w1 = oracle.or_many_heap([hs[int(ids[0])]])
        assert eng.or_many_heap(pool, ids[:1]).serialize(0) == oracle.serialize(w1)
        for h in hs + [want, ws, w1]:
            oracle.free(h)


def test_or_many_heap_tournament(engine, oracle):
    or_many_heap_body(engine, oracle)


def test_many_selection_with_a_wide_bitmap(engine, oracle):
    """or_many / xor_many through ids over a pool that holds ONE wide bitmap beside many narrow ones: the counting
    sort's rows are cut by bitmap count (a row's count field bounds the members of one key, at most one per bitmap),
    not by the pool's widest bitmap."""
    rng = np.random.default_rng(77)
    wide = (np.arange(3000, dtype=np.uint32) << np.uint32(16)) | rng.integers(0, 65536, 3000).astype(np.uint32)
    vs = [wide] + [(np.uint32(int(rng.integers(0, 3000))) << np.uint32(16)) | np.unique(rng.integers(0, 65536, 50)).astype(np.uint32)
                   for _ in range(400)]
    hs = [oracle.from_sorted(np.sort(v)) for v in vs]
    pool = engine.pool_from_serialized([oracle.serialize(h) for h in hs])
    ids = rng.permutation(len(hs)).astype(np.uint32)
    for fn, of in ((engine.or_many, oracle.or_many), (engine.xor_many, oracle.xor_many)):
        want = of([hs[i] for i in ids])
        assert fn(pool, ids).serialize(0) == oracle.serialize(want)
        oracle.free(want)
    for h in hs:
        oracle.free(h)
