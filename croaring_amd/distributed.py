"""Multi-GPU many-way aggregation: one process per GPU, torch.distributed over RCCL/xGMI.

Pairwise ops need no communication (partition the pair list).  roaring_bitmap_or_many /
xor_many over a set of bitmaps sharded across ranks needs ONE exchange step:

  1. rank r reduces ITS bitmaps to one uncompressed 1024-word chunk per distinct container key
     (Engine.many_partials -> rhip_many_partials; no cardinality, no typing);
  2. every chunk travels to the owner of its key (owner = key mod world).  RCCL has no OR/XOR
     reduction and a ring all-reduce would be bound by one xGMI link, so the exchange is a
     personalised all-to-all, which keeps all point-to-point links busy.  Two forms:
       * dense  (`key_space` given, e.g. 4096 for BASELINE config C4 where every rank sees every key):
         chunks are scattered into a zero-filled [world, key_space / world, 1024] table and exchanged
         with ONE fixed-shape `all_to_all_single` -- no count exchange, no host round trip; everything
         between stage 1 and stage 3 is stream-ordered on the engine's stream;
       * sparse (default): counts are all-gathered (one small host readback), then grouped
         point-to-point sends/receives move exactly the chunks that exist;
  3. the owner combines equal keys and canonicalises (Engine.many_finalize -> rhip_many_finalize);
     all-zero chunks of the dense form vanish there (empty results are dropped).

The result stays sharded by key (each rank holds a one-bitmap pool with the keys it owns);
`gather_serialized` collects it on one rank when a single portable bitmap is wanted.

The exchange functions only touch torch tensors, so the same code runs on CPU tensors with the gloo
backend (tests/test_distributed_cpu.py, world_size 2) and on device tensors with nccl (= RCCL); with
a gloo group and device-resident chunks (two test ranks sharing one GPU) `many_sharded` stages the
chunks through host memory.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

WORDS = 1024


def shard_ids(n_bitmaps: int, rank: int, world: int) -> np.ndarray:
    """Bitmaps b with b mod world == rank ('sharded b mod G')."""
    return np.arange(rank, n_bitmaps, world, dtype=np.uint32)


def owner_of(keys: torch.Tensor, world: int) -> torch.Tensor:
    return keys % world


def exchange_chunks(keys: torch.Tensor, words: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Sparse form: send every (key, 1024-word chunk) record to rank key % world; return what this rank owns.

    keys: int64 [n]; words: int64 [n, 1024] (same device).  Received records are in source-rank order;
    duplicates of a key (one per source rank that saw it) are combined later by many_finalize.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = keys.device
    n = keys.numel()
    words = words.reshape(n, WORDS)
    owner = owner_of(keys, world)
    order = torch.argsort(owner, stable=True)
    keys_s = keys[order].contiguous()
    words_s = words[order].contiguous()
    send_counts = torch.bincount(owner, minlength=world).to(torch.int64)
    # 1) counts: tiny all-gather (every rank learns the whole matrix)
    all_counts = [torch.zeros(world, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_counts, send_counts.to(dev), group=group)
    counts = torch.stack(all_counts).cpu()  # counts[src][dst]
    recv_counts = counts[:, rank].tolist()
    send_list = counts[rank].tolist()
    # 2) payload: grouped point-to-point (ncclGroupStart/End on RCCL; isend/irecv on gloo)
    recv_keys = torch.empty(int(sum(recv_counts)), dtype=torch.int64, device=dev)
    recv_words = torch.empty((int(sum(recv_counts)), WORDS), dtype=torch.int64, device=dev)
    ops = []
    so = np.concatenate([[0], np.cumsum(send_list)]).astype(np.int64)
    ro = np.concatenate([[0], np.cumsum(recv_counts)]).astype(np.int64)
    for peer in range(world):
        if peer == rank:
            continue
        gpeer = peer if group is None else dist.get_global_rank(group, peer)
        if send_list[peer]:
            ops.append(dist.P2POp(dist.isend, keys_s[so[peer]:so[peer + 1]], gpeer, group))
            ops.append(dist.P2POp(dist.isend, words_s[so[peer]:so[peer + 1]], gpeer, group))
        if recv_counts[peer]:
            ops.append(dist.P2POp(dist.irecv, recv_keys[ro[peer]:ro[peer + 1]], gpeer, group))
            ops.append(dist.P2POp(dist.irecv, recv_words[ro[peer]:ro[peer + 1]], gpeer, group))
    if send_list[rank]:
        recv_keys[ro[rank]:ro[rank + 1]] = keys_s[so[rank]:so[rank + 1]]
        recv_words[ro[rank]:ro[rank + 1]] = words_s[so[rank]:so[rank + 1]]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return recv_keys, recv_words


def dense_block(key_space: int, world: int) -> int:
    """Keys owned by one rank in the dense form: ceil(key_space / world)."""
    return (int(key_space) + world - 1) // world


def exchange_dense(keys: torch.Tensor, words: torch.Tensor, key_space: int, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Dense form: ONE fixed-shape all_to_all_single, no counts, no host synchronisation.

    Every key must be < key_space (the caller checks it against rhip_partials_t.max_key, a host value).  The chunk
    of key k is placed in row (k % world) * B + k // world of a zero-filled [world * B, 1024] table, B =
    ceil(key_space / world); block d of the table goes to rank d.  Returns (keys [world * B], words [world * B,
    1024]): row s * B + j is source rank s's chunk (all zero if s never saw the key) for key rank + world * j."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = keys.device
    B = dense_block(key_space, world)
    table = torch.zeros((world * B, WORDS), dtype=torch.int64, device=dev)
    if keys.numel():
        slot = (keys % world) * B + torch.div(keys, world, rounding_mode="floor")
        table.index_copy_(0, slot, words.reshape(-1, WORDS))
    recv = torch.empty_like(table)
    dist.all_to_all_single(recv, table, group=group)
    rkeys = (torch.arange(B, dtype=torch.int64, device=dev) * world + rank).repeat(world)
    return rkeys, recv


class _DevArray:
    """Zero-copy view of engine-owned device memory for torch (via __cuda_array_interface__)."""

    def __init__(self, ptr: int, shape, typestr="<i8"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}


def many_sharded(engine, local_pool, op: str = "or", ids=None, group=None, key_space: Optional[int] = None,
                 force_collective: bool = False):
    """or_many / xor_many over the union of every rank's `local_pool[ids]`.

    Returns this rank's share of the result: a one-bitmap Pool holding the container keys with
    key % world == rank.  Must be called by every rank of the group, all with the same `key_space`.
    key_space = an exclusive upper bound of the container keys on EVERY rank (e.g. 4096) selects the dense
    fixed-shape exchange; None selects the sparse one.

    Dense form: stage 1 (rhip_many_partials_dense) writes the chunks straight into the zero-filled send table and
    returns without waiting, the all_to_all_single is enqueued on the engine's stream, stage 3
    (rhip_many_finalize_dense) combines the `world` rows of every owned key -- ONE host wait, at its end."""
    dev = engine.torch_device()
    # test configuration: ranks without RCCL between them (two ranks sharing one GPU) stage chunks through the host
    staged = dist.get_backend(group) == "gloo" and dev.type != "cpu"
    if key_space is not None:
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        B = dense_block(key_space, world)
        if len(local_pool) and local_pool.max_key() >= key_space:  # (a host value, cached on the pool: no device round trip)
            raise ValueError(f"key {local_pool.max_key()} >= key_space {key_space}: use the sparse exchange")
        # torch work is issued on the ENGINE's stream: stage 1, the collective and stage 3 are ordered by the stream
        with engine.torch_stream():
            table, recv = _dense_tables(engine, world * B, dev)
            engine.many_partials_dense(op, local_pool, ids, int(key_space), world, table.data_ptr())
            if staged:
                hsend = table.cpu()
                hrecv = torch.empty_like(hsend)
                dist.all_to_all_single(hrecv, hsend, group=group)
                recv.copy_(hrecv)
            elif world == 1 and not force_collective:
                recv = table  # (a one-rank all-to-all is the identity; force_collective: issue it anyway, to time it)
            else:
                dist.all_to_all_single(recv, table, group=group)
            return engine.many_finalize_dense(op, local_pool.is64, world, rank, B, recv.data_ptr())
    parts = engine.many_partials(op, local_pool, ids)
    n = parts.n_keys
    with engine.torch_stream():
        if n:
            keys, words = engine.as_tensor(parts.d_keys, (n,)), engine.as_tensor(parts.d_words, (n, WORDS))
        else:
            keys = torch.empty(0, dtype=torch.int64, device=dev)
            words = torch.empty((0, WORDS), dtype=torch.int64, device=dev)
        if staged:
            keys, words = keys.cpu(), words.cpu()
        rk, rw = exchange_chunks(keys, words, group)
        if staged:
            rk, rw = rk.to(dev), rw.to(dev)
        rk, rw = rk.contiguous(), rw.contiguous()
        nk = rk.numel()
        out = engine.many_finalize(op, local_pool.is64, nk, rk.data_ptr() if nk else 0, rw.data_ptr() if nk else 0)
    parts.free()  # buffers go back to the context's cache; their next writer is on the same stream
    return out


def _dense_tables(engine, rows: int, dev):
    """Send / receive tables of the dense exchange, kept on the engine between calls (a fresh 32 MiB torch
    allocation per call would cost more than the exchange)."""
    cache = getattr(engine, "_dense_cache", None)
    if cache is None or cache[0].shape[0] != rows or cache[0].device != dev:
        cache = (torch.empty((rows, WORDS), dtype=torch.int64, device=dev),
                 torch.empty((rows, WORDS), dtype=torch.int64, device=dev))
        engine._dense_cache = cache
    return cache


def gather_serialized(engine, owned_pool, dst: int = 0, group=None) -> Optional[bytes]:
    """Collect the key-sharded result on rank `dst` as ONE portable-serialized bitmap.  The owned parts
    have disjoint keys, so their union is a pure pass-through copy on the device (or_many)."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    mine = owned_pool.serialize(0)
    gathered = [None] * world if rank == dst else None
    dist.gather_object(mine, gathered, dst=dst if group is None else dist.get_global_rank(group, dst), group=group)
    if rank != dst:
        return None
    is64 = owned_pool.is64
    pool = (engine.pool_from_serialized64 if is64 else engine.pool_from_serialized)(gathered)
    return engine.or_many(pool).serialize(0)
