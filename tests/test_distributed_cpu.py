"""world_size-2 gloo test of the sharded or_many exchange on CPU tensors.

The GPU stages (rhip_many_partials / rhip_many_finalize) are stood in for by the oracle here -- this
test covers what cannot be validated on one GPU: the key-owner partition, the personalised
all-to-all (counts + grouped send/recv) and that combining received chunks by key reproduces
roaring_bitmap_or_many over ALL bitmaps."""
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
