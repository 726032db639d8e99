#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Roaring set-operation engine.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   -> ONE JSON line on rank 0.

Workload (BASELINE.json configs[1], made concrete in SURVEY.md §8d "C2"): a device-resident pool of
256 bitmaps x 4096 bitset containers (8 GiB, words = splitmix64 stream, density 0.5) and the pair
schedule  k -> (k mod 256, (97 k + 1) mod 256).  One STEP = one batched roaring_bitmap_and call plus
one batched roaring_bitmap_or call over `--pairs` bitmap pairs each (default 250+250 = 500 set-ops;
the default 20 steps are the 10 000 ops of the config).  A step runs the whole hot path: key merge /
planning, the bitset x bitset kernel, result typing, directory compaction.  Inputs are resident in
HBM before the timed region; results are materialised in HBM (a result pool is recycled between
steps because 10 000 x 32 MiB of distinct outputs cannot exist at once).

With N > 1 (torchrun, one process per GPU) every rank runs the same schedule on its own pool
(pairwise ops shard with no data-path collective): scaling = "weak".

Extra keys: "roofline" (bitset x bitset kernel vs the 8 TB/s HBM peak, from HIP events on the
engine's stream) and "cpu_baseline" (CRoaring itself -- oracle/_ref -- or the C port when the
prebuilt reference is absent, on a bounded sample of the same workload on the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SEED = 0x9E3779B97F4A7C15
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BB_BYTES_PER_PAIR = 3 * 8192  # payload(a) + payload(b) + payload(result)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pool", type=int, default=256, help="bitmaps in the pool")
    ap.add_argument("--containers", type=int, default=4096, help="bitset containers per bitmap")
    ap.add_argument("--pairs", type=int, default=250, help="bitmap pairs per batched call (2 calls per step)")
    ap.add_argument("--workload", default="pairwise", choices=["pairwise", "ormany"],
                    help="pairwise = the headline C2 line (default); ormany = SURVEY C4 sharded or_many (secondary)")
    ap.add_argument("--bitmaps", type=int, default=100000, help="ormany: total sparse bitmaps over all ranks")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def schedule(k0: int, n: int, pool: int):
    k = np.arange(k0, k0 + n, dtype=np.uint64)
    return (k % pool).astype(np.uint32), ((k * 97 + 1) % pool).astype(np.uint32)


# ----------------------------------------------------------------------------- CPU baseline
def portable_bitset_bitmap(words: np.ndarray) -> bytes:
    """Portable serialization of a bitmap whose containers 0..n-1 are all bitsets."""
    import struct
    n = words.size // 1024
    cards = np.bitwise_count(words).reshape(n, 1024).sum(1).astype(np.uint32)
    assert (cards > 4096).all()
    desc = np.empty((n, 2), dtype=np.uint16)
    desc[:, 0] = np.arange(n)
    desc[:, 1] = cards - 1
    offs = (8 + 8 * n + 8192 * np.arange(n)).astype(np.uint32)
    return struct.pack("<II", 12346, n) + desc.tobytes() + offs.tobytes() + words.tobytes()


_CPU = {}


def _cpu_worker(a):
    tid, T, secs = a
    chk, hs, lhs, rhs = _CPU["chk"], _CPU["hs"], _CPU["lhs"], _CPU["rhs"]
    done, k = 0, tid
    end = time.perf_counter() + secs
    while time.perf_counter() < end:
        for op in ("and", "or"):
            r = chk.op(op, hs[lhs[k]], hs[rhs[k]])  # materialise
            chk.cardinality(r)                        # consume, as benchmarks/benchmark.cpp:2048-2057 does
            chk.free(r)
            done += 1
        k += T
    return done


def cpu_baseline(args, seconds: float):
    """Times the CPU reference (CRoaring itself when oracle/_ref is prebuilt, else the C port) on a
    bounded sample of the same workload.  CRoaring is single-threaded, so host cores are used the way
    SURVEY App. C does: the pair list is striped over T forked worker processes sharing the read-only
    inputs.  The box's memory system saturates long before all hardware threads are busy (measured:
    T=16 is the knee on the 256-thread host), so a short sweep picks the best T and reports it."""
    import multiprocessing as mp
    from gen_inputs import splitmix64
    from oracle.pyoracle import Oracle, Ref, build
    if Ref.available():
        chk, kind = Ref(), "reference"
    else:
        build()
        chk, kind = Oracle(), "port"
    n_bm = 8  # 8 x 4096 containers = 256 MiB of inputs: past the CPU caches
    hs = [chk.deserialize(portable_bitset_bitmap(splitmix64((SEED + b) & (2**64 - 1), args.containers * 1024)))
          for b in range(n_bm)]
    lhs, rhs = schedule(0, 1 << 20, n_bm)
    _CPU.update(chk=chk, hs=hs, lhs=lhs, rhs=rhs)
    ncpu = os.cpu_count() or 1
    sweep = sorted({t for t in (1, 8, 16, 32, 64) if t <= ncpu})
    per = max(1.0, seconds / len(sweep))
    best = None
    for T in sweep:
        with mp.get_context("fork").Pool(T) as pool:
            t0 = time.perf_counter()
            res = pool.map(_cpu_worker, [(t, T, per) for t in range(T)])
            dt = time.perf_counter() - t0
        rate = sum(res) / dt
        if best is None or rate > best[0]:
            best = (rate, T, sum(res), dt)
    rate, T, ops, dt = best
    one = None
    return {"value": rate, "unit": "set-ops/s", "cores": T, "kind": kind, "host_threads": ncpu,
            "sample": f"{ops} pairwise and/or ops (materialise + cardinality + free) over {n_bm} bitmaps x "
                      f"{args.containers} bitset containers in {dt:.1f} s on {T} worker processes "
                      f"(best of T={sweep}); {rate * args.containers * BB_BYTES_PER_PAIR / 1e9:.1f} GB/s algorithmic"}


# ----------------------------------------------------------------------------- C4: sharded or_many (secondary workload)
def c4_shard(n_bitmaps: int, seed: int):
    """n_bitmaps sparse bitmaps, 32 array containers each (keys stratified over [0,4096), card uniform in
    [1,512], values stratified over [0,65536)), packed back to back in portable format."""
    rng = np.random.default_rng(seed)
    NB, NK = n_bitmaps, 32
    keys = (np.arange(NK, dtype=np.uint32)[None, :] * 128 + rng.integers(0, 128, (NB, NK), dtype=np.uint32))
    cards = rng.integers(1, 513, (NB, NK), dtype=np.uint32)
    ccum = np.concatenate([[0], np.cumsum(cards.ravel(), dtype=np.int64)])
    total = int(ccum[-1])
    cid = np.repeat(np.arange(NB * NK, dtype=np.int64), cards.ravel())
    within = np.arange(total, dtype=np.int64) - ccum[cid]
    stride = (65536 // cards.ravel().astype(np.int64))[cid]
    vals = (within * stride + (rng.integers(0, 1 << 30, total, dtype=np.int64) % stride)).astype(np.uint16)
    hdr = 8 + 8 * NK
    per_bm = cards.sum(1).astype(np.int64) * 2 + hdr
    offs = np.concatenate([[0], np.cumsum(per_bm)]).astype(np.int64)
    blob = np.zeros(int(offs[-1]), dtype=np.uint8)
    h32 = np.zeros((NB, hdr // 4), dtype=np.uint32)
    h32[:, 0] = 12346
    h32[:, 1] = NK
    h32[:, 2:2 + NK] = keys | ((cards - 1) << 16)
    inner = np.concatenate([np.zeros((NB, 1), np.int64), np.cumsum(cards.astype(np.int64) * 2, 1)[:, :-1]], 1) + hdr
    h32[:, 2 + NK:] = inner.astype(np.uint32)
    blob[(offs[:-1, None] + np.arange(hdr)[None, :]).ravel()] = h32.view(np.uint8).ravel()
    vstart = np.concatenate([[0], np.cumsum(cards.sum(1).astype(np.int64))])
    bm = cid // NK
    pos = offs[bm] + hdr + 2 * (np.arange(total, dtype=np.int64) - vstart[bm])
    blob[pos] = (vals & 0xFF).astype(np.uint8)
    blob[pos + 1] = (vals >> 8).astype(np.uint8)
    return blob, offs[:-1], per_bm


def run_ormany(args, eng, rank, world, barrier):
    """Strong scaling: --bitmaps sparse bitmaps in total, rank r holds bitmaps/world of them; one step = one
    or_many over ALL of them = per-rank partial chunks -> key-owner exchange over RCCL -> owner finalize."""
    import torch.distributed as dist
    from croaring_amd.distributed import many_sharded
    n_local = args.bitmaps // world
    blob, offs, lens = c4_shard(n_local, 4 + 1000 * rank + world)
    pool = eng.pool_from_packed(blob, offs, lens)
    payload = pool.payload_bytes()

    def step():
        return many_sharded(eng, pool, "or") if world > 1 else eng.or_many(pool)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    tot_payload = payload
    if world > 1:
        import torch
        t = torch.tensor([dt, float(payload)], dtype=torch.float64, device="cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, tot_payload = float(tmax[0].item()), float(t[1].item())
    return {
        "metric": "set-ops/sec + GB/s (pairwise AND/OR; or_many) on realdata, 1/2/4/8 GPU",
        "value": args.steps / dt, "unit": "or_many-ops/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": f"C4 or_many over {n_local * world} sparse bitmaps x 32 array containers, "
                               f"sharded {n_local} per GPU, key-owner exchange over RCCL",
                   "algorithmic_GBps": tot_payload * args.steps / dt / 1e9,
                   "result_cardinality_rank0": int(out.cardinalities()[0])},
    }


# ----------------------------------------------------------------------------- main
def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch  # first: the engine then binds to the HIP runtime torch already loaded
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import croaring_amd
    eng = croaring_amd.Engine(local_rank)
    eng.set_timing(True)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload == "ormany":
        out = run_ormany(args, eng, rank, world, barrier)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    pool = eng.pool_synth_bitset(args.pool, args.containers, SEED + 1000003 * rank)

    results = {"and": None, "or": None}
    bb_ms, bb_pairs = [], []

    def step(i: int, timed: bool):
        for j, op in enumerate(("and", "or")):
            lhs, rhs = schedule((2 * i + j) * args.pairs, args.pairs, args.pool)
            results[op] = eng.pairwise(op, pool, lhs, pool, rhs, reuse=results[op])
            if timed:
                st = eng.last_stats()
                bb_ms.append(st["ms_bitset_kernel"])
                bb_pairs.append(st["n_bitset_pairs"])

    for i in range(args.warmup):
        step(i, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i, True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity: every result container of the last OR batch is a bitset, cardinalities are plausible
    assert results["or"].type_counts() == (args.pairs * args.containers, 0, 0)

    ops_per_step = 2 * args.pairs
    total_ops = ops_per_step * args.steps * world
    ms_kernel = float(np.mean(bb_ms)) if bb_ms else 0.0
    pairs_per_launch = float(np.mean(bb_pairs)) if bb_pairs else 0.0
    achieved = (pairs_per_launch * BB_BYTES_PER_PAIR) / (ms_kernel * 1e-3) / 1e9 if ms_kernel > 0 else 0.0
    traffic = None
    tp = os.path.join(ROOT, "profiles", "bb_traffic.json")
    if os.path.exists(tp):
        try:
            traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "set-ops/sec + GB/s (pairwise AND/OR; or_many) on realdata, 1/2/4/8 GPU",
        "value": total_ops / dt,
        "unit": "set-ops/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": f"C2 synthetic bitset-only: pool {args.pool} bitmaps x {args.containers} bitset "
                               f"containers (density 0.5), batched pairwise AND+OR, {args.pairs} pairs per call",
                   "ops_per_step": ops_per_step,
                   "algorithmic_GBps": total_ops * args.containers * BB_BYTES_PER_PAIR / dt / 1e9,
                   "parallelism": f"pairs sharded over {world} GPU(s), no data-path collective"},
        "roofline": {"bound": "hbm", "kernel": "k_bb (bitset x bitset fused op+popcount)", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "avg_launch_ms": ms_kernel, "pairs_per_launch": pairs_per_launch},
    }
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
