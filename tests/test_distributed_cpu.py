"""world_size-2 gloo tests of the sharded or_many / xor_many without a GPU.

* test_sharded_many_composed_gloo: croaring_amd.distributed.many_sharded END TO END with two ranks -- the real
  rhip_many_partials and rhip_many_finalize (the kernel sources running under the tests/emu SIMT emulator, whose
  "device" memory torch sees as CPU tensors), both exchange forms (sparse: counts + grouped send/recv; dense:
  one fixed-shape all_to_all_single), or and xor, checked against the oracle's or_many / xor_many over ALL bitmaps.
* test_sharded_or_many_exchange_gloo: the exchange alone with the oracle standing in for both stages (kept: it
  isolates a routing bug from a kernel bug).
* test_sharded_many64_gloo: the same end-to-end path on ROARING64 pools (48-bit container keys, owner = key mod
  world, sparse exchange): seeded random 64-bit bitmaps and a slice of BASELINE config C5 (wikileaks-noquotes x 10
  high-32 buckets), checked against the oracle's roaring64 or / xor folds.
The same composed path runs on a real MI355X in tests/test_gpu_distributed.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _chunks_of(oracle, hs):
    """CPU stand-in for rhip_many_partials: one uncompressed chunk per distinct key of the shard."""
    u = oracle.or_many(hs)
    vals = oracle.to_array(u)
    oracle.free(u)
    keys = np.unique(vals >> 16).astype(np.int64)
    words = np.zeros((len(keys), 1024), dtype=np.uint64)
    idx = np.searchsorted(keys, (vals >> 16).astype(np.int64))
    low = vals & 0xFFFF
    np.bitwise_or.at(words, (idx, (low >> 6).astype(np.int64)), np.uint64(1) << (low & 63).astype(np.uint64))
    return keys, words.view(np.int64)


def _worker(rank, world, port, xor, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from croaring_amd.distributed import exchange_chunks, shard_ids
    from gen_inputs import random_bitmap
    from oracle.pyoracle import Oracle
    oracle = Oracle()
    rng = np.random.default_rng(99)  # same stream on every rank: identical global input set
    allv = [random_bitmap(rng, max_keys=8, key_space=16) for _ in range(21)]
    mine = [oracle.from_sorted(allv[i]) for i in shard_ids(len(allv), rank, world)]
    keys, words = _chunks_of(oracle, mine)
    rk, rw = exchange_chunks(torch.from_numpy(keys), torch.from_numpy(words))
    rk, rw = rk.numpy(), rw.numpy().view(np.uint64)
    assert np.all(rk % world == rank), "received a key this rank does not own"
    # owner-side combine (stand-in for rhip_many_finalize)
    ukeys = np.unique(rk)
    acc = np.zeros((len(ukeys), 1024), dtype=np.uint64)
    np.bitwise_or.at(acc, np.searchsorted(ukeys, rk), rw)
    bits = np.unpackbits(acc.view(np.uint8), bitorder="little").reshape(len(ukeys), 65536)
    kk, low = np.nonzero(bits)
    got = ((ukeys[kk].astype(np.uint64) << np.uint64(16)) | low.astype(np.uint64)).astype(np.uint32)
    # expected: or_many over ALL bitmaps, restricted to owned keys
    hs_all = [oracle.from_sorted(v) for v in allv]
    full = oracle.to_array(oracle.or_many(hs_all))
    want = full[((full >> 16) % world) == rank]
    q.put((rank, bool(np.array_equal(np.sort(got), want)), int(len(want))))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_or_many_exchange_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, False, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
    assert sum(r[2] for r in res) > 0


def test_shard_ids_partition():
    sys.path.insert(0, ROOT)
    from croaring_amd.distributed import shard_ids
    for world in (1, 2, 4, 8):
        parts = [shard_ids(1000, r, world) for r in range(world)]
        assert np.array_equal(np.sort(np.concatenate(parts)), np.arange(1000))


def _composed_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from croaring_amd.distributed import gather_serialized, many_sharded, shard_ids
        from emu import emu_engine
        from gen_inputs import random_bitmap
        from oracle.pyoracle import Oracle
        oracle = Oracle()
        eng = emu_engine()
        rng = np.random.default_rng(4242)  # same stream on every rank: identical global input set
        allv = [random_bitmap(rng, max_keys=8, key_space=16) for _ in range(23)]
        hs_all = [oracle.from_sorted(v) for v in allv]
        mine = [int(i) for i in shard_ids(len(allv), rank, world)]
        pool = eng.pool_from_serialized([oracle.serialize(hs_all[i]) for i in mine])
        ok = []
        for op, fn in (("or", oracle.or_many), ("xor", oracle.xor_many)):
            want = oracle.to_array(fn(hs_all))
            for key_space in (None, 16, 21):
                owned = many_sharded(eng, pool, op, key_space=key_space)
                hv = oracle.deserialize(owned.serialize(0))
                v = oracle.to_array(hv)
                ok.append(bool(oracle.validate(hv)) and bool(np.all((v >> 16) % world == rank))
                          and bool(np.array_equal(v, want[((want >> 16) % world) == rank])))
                blob = gather_serialized(eng, owned)
                if rank == 0:
                    hg = oracle.deserialize(blob)
                    ok.append(bool(np.array_equal(oracle.to_array(hg), want)))
        try:
            many_sharded(eng, pool, "or", key_space=3)  # a key >= key_space must be refused, on every rank alike
            ok.append(False)
        except ValueError:
            ok.append(True)
        q.put((rank, all(ok), len(ok)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_many_composed_gloo():
    from emu import build_emu
    if not os.path.exists(build_emu.CXX):
        pytest.skip("hipemu needs the ROCm clang++")
    build_emu.build()
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_composed_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res


def _dense_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from croaring_amd.distributed import dense_block, exchange_dense
    # rank r holds keys {k : k % 3 != r} of [0, 11), chunk content = key * 1000 + source rank
    ks = 11
    keys = torch.tensor([k for k in range(ks) if k % 3 != rank], dtype=torch.int64)
    words = (keys * 1000 + rank).reshape(-1, 1).repeat(1, 1024)
    rk, rw = exchange_dense(keys, words, ks)
    B = dense_block(ks, world)
    ok = rk.numel() == world * B and rw.shape == (world * B, 1024)
    for s_ in range(world):
        for j in range(B):
            k = rank + world * j
            row = rw[s_ * B + j]
            ok = ok and int(rk[s_ * B + j]) == k
            has = k < ks and k % 3 != s_
            ok = ok and bool((row == (k * 1000 + s_ if has else 0)).all())
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_dense_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dense_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def _rand64(rng, random_bitmap):
    highs = rng.choice(6, int(rng.integers(0, 4)), replace=False)
    parts = [(np.uint64(int(h) * 7 + 1) << np.uint64(32)) | random_bitmap(rng, max_keys=4, key_space=6).astype(np.uint64)
             for h in highs]
    return np.sort(np.concatenate(parts)) if parts else np.zeros(0, np.uint64)


def _many64_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from croaring_amd.distributed import gather_serialized, many_sharded, shard_ids
        from emu import emu_engine
        from gen_inputs import random_bitmap
        from oracle.pyoracle import Oracle
        from util import c5_inputs, load_bundle
        oracle = Oracle()
        eng = emu_engine()
        rng = np.random.default_rng(6464)  # same stream on every rank: identical global input set
        sets = {"rand64": [oracle.serialize64(oracle.from_sorted64(_rand64(rng, random_bitmap))) for _ in range(19)],
                "c5_slice": c5_inputs(load_bundle("wikileaks-noquotes")[:24])}
        ok = []
        for name, bufs in sets.items():
            hs_all = [oracle.deserialize64(b) for b in bufs]
            mine = [int(i) for i in shard_ids(len(bufs), rank, world)]
            pool = eng.pool_from_serialized64([bufs[i] for i in mine])
            assert pool.is64
            want_or = oracle.or_many64(hs_all)
            want_xor = oracle.deserialize64(bufs[0])
            for h in hs_all[1:]:
                nx = oracle.op64("xor", want_xor, h)
                oracle.free64(want_xor)
                want_xor = nx
            for op, want in (("or", want_or), ("xor", want_xor)):
                owned = many_sharded(eng, pool, op)  # 48-bit keys: the sparse exchange
                # every container key this rank ends up with is one it owns
                vals, _ = owned.to_values()
                ok.append(owned.is64 and bool(np.all(((vals >> np.uint64(16)) % np.uint64(world)) == np.uint64(rank))))
                blob = gather_serialized(eng, owned)
                if rank == 0:
                    hg = oracle.deserialize64(blob)
                    x = oracle.op64("xor", hg, want)
                    ok.append(oracle.cardinality64(x) == 0 and oracle.cardinality64(hg) == oracle.cardinality64(want))
                    oracle.free64(x)
                    oracle.free64(hg)
            try:
                many_sharded(eng, pool, "or", key_space=4096)  # 48-bit keys do not fit a dense table: refused
                ok.append(pool.max_key() < 4096)
            except ValueError:
                ok.append(True)
        q.put((rank, all(ok), len(ok)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_many64_gloo():
    """BASELINE configs[4] / SURVEY §8e "64-bit: identical, owner = key mod G": roaring64 pools sharded over two ranks,
    the real stage-1 / stage-3 entry points through the kernel emulator."""
    from emu import build_emu
    if not os.path.exists(build_emu.CXX):
        pytest.skip("hipemu needs the ROCm clang++")
    build_emu.build()
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_many64_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res), res
