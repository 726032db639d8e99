"""Multi-GPU many-way aggregation: one process per GPU, torch.distributed over RCCL/xGMI.

Pairwise ops need no communication (partition the pair list).  roaring_bitmap_or_many /
xor_many over a set of bitmaps sharded across ranks needs ONE exchange step:

  1. rank r reduces ITS bitmaps to one uncompressed 1024-word chunk per distinct container key
     (Engine.many_partials -> rhip_many_partials; no cardinality, no typing);
  2. every chunk travels to the owner of its key (owner = key mod world) -- a personalised
     all-to-all built from grouped point-to-point sends/receives, which is what maps onto xGMI's
     point-to-point links (RCCL has no OR/XOR reduction, SURVEY G10, and a ring all-reduce would be
     bound by a single link);
  3. the owner combines equal keys and canonicalises (Engine.many_finalize -> rhip_many_finalize).

The result stays sharded by key (each rank holds a one-bitmap pool with the keys it owns);
`gather_serialized` collects it on one rank when a single portable bitmap is wanted.

`exchange_chunks` only touches torch tensors, so the same code runs on CPU tensors with the gloo
backend (tests/test_distributed_cpu.py, world_size 2) and on device tensors with nccl (= RCCL).
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
    """Send every (key, 1024-word chunk) record to rank key % world; return what this rank owns.

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


class _DevArray:
    """Zero-copy view of engine-owned device memory for torch (via __cuda_array_interface__)."""

    def __init__(self, ptr: int, shape, typestr="<i8"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}


def many_sharded(engine, local_pool, op: str = "or", ids=None, group=None):
    """or_many / xor_many over the union of every rank's `local_pool[ids]`.

    Returns this rank's share of the result: a one-bitmap Pool holding the container keys with
    key % world == rank.  Must be called by every rank of the group.
    """
    parts = engine.many_partials(op, local_pool, ids)
    n = parts.n_keys
    dev = torch.device("cuda", torch.cuda.current_device())
    if n:
        keys = torch.as_tensor(_DevArray(parts.d_keys, (n,)), device=dev)
        words = torch.as_tensor(_DevArray(parts.d_words, (n, WORDS)), device=dev)
    else:
        keys = torch.empty(0, dtype=torch.int64, device=dev)
        words = torch.empty((0, WORDS), dtype=torch.int64, device=dev)
    engine.synchronize()
    rk, rw = exchange_chunks(keys, words, group)
    torch.cuda.synchronize()
    out = engine.many_finalize(op, local_pool.is64, rk.numel(), rk.data_ptr() if rk.numel() else 0,
                               rw.data_ptr() if rk.numel() else 0)
    parts.free()
    return out


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
