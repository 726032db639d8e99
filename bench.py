#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Roaring set-operation engine.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   -> ONE JSON line on rank 0.
With N > 1 and no torchrun environment the script re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`; it refuses to print a
line whose n_gpus differs from --gpus.

Headline workload (BASELINE.json configs[1], made concrete in SURVEY.md §8d "C2"): a device-resident pool of
256 bitmaps x 4096 bitset containers (8 GiB, words = splitmix64 stream, density 0.5) and the pair schedule
k -> (k mod 256, (97 k + 1) mod 256).  One STEP = `--rounds` x (one batched roaring_bitmap_and call + one batched
roaring_bitmap_or call over `--pairs` bitmap pairs each): default 6 x (250 + 250) = 3000 set-ops, so the default
24 steps time > 1 s (49 ms per step).  A step runs the whole hot path: key merge / planning, the bitset x bitset kernel, result
typing, directory compaction.  Inputs are resident in HBM before the timed region; results are materialised in HBM
(result pools are recycled between calls: 60 000 x 32 MiB of distinct outputs cannot exist at once).
N > 1: every rank runs the same schedule on its own pool (pairwise ops shard with no data-path collective,
SURVEY §8e): scaling = "weak".

config.secondary_summary (same line, measured after the timed region of the headline; one compact row per configuration
-- [ms per batch, fraction of the HBM peak, checksum ok, ...] -- so that the line stays ~4 KB; every other figure of a row
goes to stderr as `BENCH_DETAIL {...}` and to gpurun_out/bench_detail.json):
  c3_* / c1_*   realdata weather_sept_85 / census1881 (tests/golden bundles): ALL unordered pairs, one batched call
                per op (and / or / xor / andnot / and_cardinality) over a PREPARED pair list (rhip_pairlist_*), wall time
                of the whole call, calls issued back to back; the SURVEY §8d checksums are asserted; pairs are
                partitioned over ranks (strong scaling); CRoaring on 1 host core beside it.
  c4_or_many    BASELINE configs[3]: roaring_bitmap_or_many over 100 000 seeded sparse bitmaps (pcg32 generator in
                the library), bitmaps b mod N on rank b mod N, key-owner exchange over RCCL (strong scaling).
  c4x10_or_many the same generator at 10^6 bitmaps (16.4 GB): the many-way row large enough for N GPUs to split;
                cardinality checked against the reference's (tests/golden/c4x10_or_many.npz).
  c4_shard_stages  (N = 1 only) what ONE rank of 2 / 4 / 8 pays for the sharded or_many, measured on this GPU.
  c5_*          BASELINE configs[4]: roaring64, wikileaks-noquotes x 10 buckets, all pairs and/or + the 200-way union
                (N > 1: sharded b mod N through the 48-bit-key exchange, cardinality asserted).

Extra keys: "roofline" (bitset x bitset kernel vs the 8 TB/s HBM peak, from HIP events on the engine's stream; the two
result arenas are placed by the LIBRARY when it allocates them -- config.result_arena_placement lists the probe rates)
and "cpu_baseline" (CRoaring itself -- oracle/_ref -- or the C port when the prebuilt reference is absent, on a bounded
sample of the same workload on rank 0's host cores, at every N: 1 core, best T of a sweep, the ISA variants).
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
METRIC = "set-ops/sec + GB/s (pairwise AND/OR; or_many) on realdata, 1/2/4/8 GPU"
# SURVEY §8d checksums (sum over all unordered pairs of the result cardinality), asserted inside the bench
CHECKSUMS = {
    "weather_sept_85": {"and": 24220711, "or": 1232335437, "xor": 1208114726, "andnot": 581541349},
    "census1881": {"and": 15213, "or": 199753126, "xor": 199737913, "andnot": 83195751},
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--arena-tries", type=int, default=0,
                    help="diagnostic: caller-side start-up allocations of result pools, the two fastest kept (Engine.pairwise_placed). "
                         "0 (default): nothing -- the library places a large result arena by measurement itself (RHIP_ARENA_TRIES)")
    ap.add_argument("--pool", type=int, default=256, help="bitmaps in the pool")
    ap.add_argument("--containers", type=int, default=4096, help="bitset containers per bitmap")
    ap.add_argument("--pairs", type=int, default=250, help="bitmap pairs per batched call")
    ap.add_argument("--rounds", type=int, default=6, help="(AND call + OR call) rounds per step")
    ap.add_argument("--workload", default="pairwise", choices=["pairwise", "ormany"],
                    help="pairwise = the headline C2 line (default); ormany = C4 sharded or_many as the headline")
    ap.add_argument("--bitmaps", type=int, default=100000, help="C4: total sparse bitmaps over all ranks")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the stored PMC pass (profiles/bb_traffic.json) instead of two rocprofv3 --pmc child runs")
    ap.add_argument("--no-secondary", action="store_true", help="skip config.secondary (realdata, C4, C5)")
    ap.add_argument("--no-x10", action="store_true", help="skip the 10^6-bitmap or_many row (16 GB of images built on the host, ~20 s)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0)
    ap.add_argument("--reps", type=int, default=20, help="repetitions of every secondary batch (>= 20 in a measurement run)")
    ap.add_argument("--ranks-share-device", action="store_true",
                    help="DRY RUN of the N > 1 line on ONE GPU: every rank opens cuda:0, the process group is gloo and the "
                         "many-way exchange is staged through host memory (croaring_amd.distributed).  Every statement of the "
                         "multi-rank script runs except the collectives on RCCL; the line says so and is not a scaling number")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="internal: run the cpu_baseline leg alone and print its JSON (rank 0 does this in a fresh process at N > 1)")
    return ap.parse_args()


def schedule(k0: int, n: int, pool: int):
    k = np.arange(k0, k0 + n, dtype=np.uint64)
    return (k % pool).astype(np.uint32), ((k * 97 + 1) % pool).astype(np.uint32)


# ----------------------------------------------------------------------------- launch
def maybe_spawn(args):
    """--gpus N > 1 without a torchrun environment: become `torch.distributed.run` with N ranks."""
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)


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
    done, k, n = 0, tid, len(lhs)
    end = time.perf_counter() + secs
    while time.perf_counter() < end:
        k %= n  # (small containers: a worker outruns the 2^20-pair schedule within its window)
        for op in ("and", "or"):
            r = chk.op(op, hs[lhs[k]], hs[rhs[k]])  # materialise
            chk.cardinality(r)                        # consume, as benchmarks/benchmark.cpp:2048-2057 does
            chk.free(r)
            done += 1
        k += T
    return done


def _one_core(chk, hs, lhs, rhs, min_reps=10, min_s=0.2):
    """SURVEY §8d timing rule (the reference harness's own: >= 10 repetitions and >= 200 ms): one repetition = one
    and + one or (materialise + cardinality + free); returns (min, median) seconds per OP."""
    ts, k, t_all = [], 0, time.perf_counter()
    while len(ts) < min_reps or time.perf_counter() - t_all < min_s:
        t0 = time.perf_counter()
        for op in ("and", "or"):
            r = chk.op(op, hs[lhs[k]], hs[rhs[k]])
            chk.cardinality(r)
            chk.free(r)
        ts.append((time.perf_counter() - t0) / 2)
        k = (k + 1) % len(lhs)
    return float(np.min(ts)), float(np.median(ts))


def _ref_variants():
    from oracle.pyoracle import Ref
    out = []
    for label, fname in (("avx512", "libcroaring_ref.so"), ("avx2", "libcroaring_ref_noavx512.so"),
                         ("scalar", "libcroaring_ref_noavx.so")):
        path = os.path.join(ROOT, "oracle", "_ref", fname)
        if os.path.exists(path):
            out.append((label, type("RefVariant", (Ref,), {"PATH": path})))
    return out


def live_bb_traffic(args):
    """HBM bytes per k_bb launch, MEASURED NOW: two child runs of this script's headline under `rocprofv3 --pmc` -- FETCH_SIZE
    and WRITE_SIZE each in its own pass, with --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes (units of
    KiB; FETCH_SIZE reports half of the bytes of a wide coalesced read on gfx950 and is doubled; both are calibrated in the
    same pass on k_synth_fill / k_synth_dir, which write / read the pool exactly once).  ~25 s per pass; any failure returns
    (None, reason) and the caller falls back to the stored figure."""
    import csv, glob, shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    pool_bytes = args.pool * args.containers * 8192
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="rhip_pmc_", dir="/tmp")
        try:
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "b", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--rounds", "1", "--no-cpu", "--no-secondary",
                   "--no-live-traffic", "--pool", str(args.pool), "--containers", str(args.containers), "--pairs", str(args.pairs)]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
            env["TMPDIR"] = "/tmp"
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd="/tmp")
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} exited {p.returncode}: {p.stderr[-200:]!r}"
            agg = {}
            for r in csv.DictReader(open(files[0])):
                if r["Counter_Name"] == counter:
                    agg.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), []).append(float(r["Counter_Value"]))
            bb = [v for k, vs in agg.items() if k.startswith("k_bb<") for v in vs]
            cal = agg.get("k_synth_fill" if counter == "WRITE_SIZE" else "k_synth_dir", [])
            if not bb or not cal:
                return None, f"no k_bb / calibration rows in the {counter} pass"
            res[counter] = (sum(bb) / len(bb) * 1024.0, cal[0] * 1024.0 / pool_bytes, len(bb))
        except Exception as e:
            return None, f"{counter} pass: {str(e)[:160]}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    rd, cal_r, n_r = res["FETCH_SIZE"]
    wr, cal_w, n_w = res["WRITE_SIZE"]
    return {"hbm_bytes_per_launch": 2.0 * rd + wr, "read_bytes": 2.0 * rd, "write_bytes": wr, "launches": [n_r, n_w],
            "calibration": {"k_synth_dir FETCH_SIZE x1024 / pool bytes (0.5 = the gfx950 half count)": round(cal_r, 4),
                            "k_synth_fill WRITE_SIZE x1024 / pool bytes": round(cal_w, 4)}}, None


def cpu_baseline(args, seconds: float):
    """Times the CPU reference (CRoaring itself when oracle/_ref is prebuilt, else the C port) on a bounded sample of
    the headline workload: 8 bitmaps x 4096 bitset containers (256 MiB of inputs: past the CPU caches), the same
    pair schedule, and + or per repetition, each op materialised + cardinality + free.  CRoaring is single-threaded,
    so "cores" means forked worker processes striping the pair list over shared read-only inputs.
    Reported: 1 core (min / median of >= 10 reps), all hardware threads, and the best of a short sweep (the host's
    memory system saturates long before all threads are busy); `value` is the best."""
    import multiprocessing as mp
    from gen_inputs import splitmix64
    from oracle.pyoracle import Oracle, Ref, build
    if Ref.available():
        chk, kind = Ref(), "reference"
    else:
        build()
        chk, kind = Oracle(), "port"
    n_bm = 8
    bufs = [portable_bitset_bitmap(splitmix64((SEED + b) & (2**64 - 1), args.containers * 1024)) for b in range(n_bm)]
    hs = [chk.deserialize(b) for b in bufs]
    lhs, rhs = schedule(0, 1 << 20, n_bm)
    _CPU.update(chk=chk, hs=hs, lhs=lhs, rhs=rhs)
    ncpu = os.cpu_count() or 1
    gb = args.containers * BB_BYTES_PER_PAIR / 1e9
    tmin, tmed = _one_core(chk, hs, lhs, rhs)
    one = {"ops_per_s_best": 1 / tmin, "ops_per_s_median": 1 / tmed, "GBps_median": gb / tmed}
    sweep = sorted({t for t in (8, 16, 24, 32, 48, 64, ncpu) if 1 < t <= ncpu})
    per = max(1.0, (seconds - 2.0) / max(1, len(sweep) + 1))
    multi = {}
    for T in sweep:
        windows = 1
        rates = []
        for _ in range(windows):
            with mp.get_context("fork").Pool(T) as pool:
                t0 = time.perf_counter()
                res = pool.map(_cpu_worker, [(t, T, per / windows) for t in range(T)])
                dt = time.perf_counter() - t0
            rates.append(sum(res) / dt)
        multi[T] = {"ops_per_s_median": float(np.median(rates)), "ops_per_s_min": float(np.min(rates))}
    best_T, best = 1, one["ops_per_s_median"]
    for T, r in multi.items():
        if r["ops_per_s_median"] > best:
            best_T, best = T, r["ops_per_s_median"]
    isa = {}
    for label, cls in _ref_variants():
        try:
            v = cls()
            vh = [v.deserialize(b) for b in bufs[:4]]
            a, m = _one_core(v, vh, lhs % 4, rhs % 4, min_reps=10, min_s=0.2)
            isa[label] = {"ops_per_s_median_1core": 1 / m, "GBps_median": gb / m}
            for h in vh:
                v.free(h)
        except Exception as e:  # a variant that does not load on this host is reported, not fatal
            isa[label] = {"error": str(e)[:80]}
    for h in hs:
        chk.free(h)
    return {"value": best, "unit": "set-ops/s", "cores": best_T, "kind": kind, "host_threads": ncpu,
            "one_core": one, "all_cores": dict(multi.get(ncpu, {}), procs=ncpu) if ncpu in multi else None,
            "sweep": {str(T): r for T, r in multi.items()}, "isa_1core": isa or None,
            "sample": f"pairwise and+or (materialise + cardinality + free) over {n_bm} bitmaps x {args.containers} "
                      f"bitset containers, schedule k -> (k mod 8, (97k+1) mod 8); 1 core: >= 10 reps and >= 200 ms, "
                      f"min/median; T worker processes for {per:.1f} s per T in {sweep}; best = T={best_T}: "
                      f"{best * gb:.1f} GB/s algorithmic"}


# ----------------------------------------------------------------------------- secondary workloads
class Dist:
    """The little bit of torch.distributed the bench needs, degenerate at world 1."""

    def __init__(self, rank, world, torch, dist, reps=20, host_tensors=False):
        self.rank, self.world, self.torch, self.dist = rank, world, torch, dist
        self.reps = reps
        self.rdev = "cpu" if host_tensors else "cuda"  # where the scalars of max / sum live (gloo: host)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max(self, x: float) -> float:
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.rdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, x: float) -> float:
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.rdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())


def timed_calls(D: Dist, fn, reps=None, min_s=0.05):
    """>= reps repetitions (and >= min_s in total) of fn; returns (min, median) of the wall time of a call.  fn is a
    SYNCHRONOUS call (it returns when its results are complete), so on one GPU the calls are simply issued one after the
    other, as a caller would; with several ranks every call is bracketed by a barrier and the slowest rank counts."""
    fn()
    fn()
    sync = D.world > 1
    reps = D.reps if reps is None else reps
    if reps < 20:
        min_s = 0.0  # (a dry run: the statements, not the statistics)
    ts, t_all = [], time.perf_counter()
    while len(ts) < reps or time.perf_counter() - t_all < min_s:
        if sync:
            D.barrier()
        t0 = time.perf_counter()
        fn()
        if sync:
            D.barrier()
        dt = time.perf_counter() - t0
        ts.append(D.max(dt) if sync else dt)
        if len(ts) >= 200:
            break
    return float(np.min(ts)), float(np.median(ts))


def cpu_pairs_rate(chk, hs, lhs, rhs, op, budget=0.4, is64=False):
    """CRoaring on one host core over the same pair list (bounded to `budget` seconds), ops/s."""
    fop, fcard, ffree = (chk.op64, chk.cardinality64, chk.free64) if is64 else (chk.op, chk.cardinality, chk.free)
    n, t0 = 0, time.perf_counter()
    for i, j in zip(lhs, rhs):
        r = fop(op, hs[i], hs[j])
        fcard(r)
        ffree(r)
        n += 1
        if (n & 255) == 0 and time.perf_counter() - t0 > budget:
            break
    return n / (time.perf_counter() - t0)


def _stored_traffic():
    """profiles/realdata_traffic.json: HBM bytes per all-pairs batch and issue shares of its kernels, from rocprofv3 PMC
    passes recorded earlier (scripts/gpu_final_r5.sh + scripts/summarize_realdata_traffic.py) -- NOT re-measured in this run."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "realdata_traffic.json")))
    except Exception:
        return {}


def run_realdata(eng, D: Dist, name: str, tag: str, chk, is64=False, ops=("and", "or", "xor", "andnot")):
    from util import all_pairs, c5_inputs, load_bundle
    traffic = _stored_traffic() if D.world == 1 else {}
    if is64:
        bufs = c5_inputs()
        pool = eng.pool_from_serialized64(bufs)
    else:
        bufs = load_bundle(name)
        pool = eng.pool_from_serialized(bufs)
    L, R = all_pairs(len(bufs))
    lhs, rhs = L[D.rank::D.world].copy(), R[D.rank::D.world].copy()
    hs = None
    if chk is not None and D.rank == 0:
        hs = [(chk.deserialize64 if is64 else chk.deserialize)(b) for b in bufs]
    out = {}
    # The pair list is prepared once (rhip_pairlist_*: validated, summed, resident on the device), as a caller that walks
    # the same pairs with every op -- the reference benchmark's loops -- would; `ms_adhoc_list` is the same call handed
    # the two index arrays from the host every time (rhip_pairwise).
    plist = eng.pairlist_all_pairs(pool) if D.world == 1 else eng.pairlist(pool, lhs, pool, rhs)
    for op in ops:
        res = [None]

        def call():
            res[0] = eng.pairwise_list(op, plist, reuse=res[0])

        def call_adhoc():
            res[0] = eng.pairwise(op, pool, lhs, pool, rhs, reuse=res[0])
        def call_first():  # a batch that has to PLAN: the list's cached plans dropped in front of it (untimed)
            plist.drop_plans()
            t0 = time.perf_counter()
            call()
            return time.perf_counter() - t0
        amin, amed = timed_calls(D, call_adhoc)
        call_first()
        fmed = float(np.median([D.max(call_first()) for _ in range(max(5, min(D.reps, 15)))]))
        tmin, tmed = timed_calls(D, call)  # (every call after the first takes its plan from the list: rhip_pairlist, round 6)
        st = eng.last_stats()
        alg = D.sum(float(st["bytes_in"] + st["bytes_out"]))
        csum = D.sum(float(res[0].cardinalities().sum()))
        want = CHECKSUMS.get(name, {}).get(op)
        if want is not None and D.rank == 0:
            assert int(csum) == want, f"{name} {op}: checksum {int(csum)} != SURVEY §8d {want}"
        row = {"pairs": int(L.size), "ops_per_s": L.size / tmed, "ops_per_s_best": L.size / tmin,
               "ms_batch_median": tmed * 1e3, "ms_batch_min": tmin * 1e3, "ms_first_call": fmed * 1e3,
               "plan_cached": bool(eng.plan_cached()), "ms_adhoc_list": amed * 1e3, "alg_GBps": alg / tmed / 1e9,
               "frac_first_call": alg / fmed / 1e9 / HBM_PEAK_GBS,
               "frac": alg / tmed / 1e9 / HBM_PEAK_GBS, "checksum": int(csum), "checksum_ok": want is None or int(csum) == want,
               "matched_pairs_rank0": int(st["matched_pairs"]), "passthrough_rank0": int(st["passthrough"])}
        # `frac` is ALGORITHMIC bytes over time -- on a 7.6 MB data set the operands live in the L2s, so it is
        # not HBM utilisation.  The stored PMC pass says what really crossed the memory side per batch and what the batch's
        # kernels spent their wave cycles on: that names the bound of the row.
        tr = traffic.get(f"{tag}_{op}")
        if tr:
            hb = tr["hbm_read_bytes"] + tr["hbm_write_bytes"]
            row["hbm_traffic_frac"] = hb / tmed / 1e9 / HBM_PEAK_GBS
            row["hbm_traffic_over_algorithmic"] = hb / max(1.0, alg)
            row["valu_share_of_wave_cycles"] = tr.get("valu_share_of_wave_cycles")
            row["lds_bank_conflict_share"] = tr.get("lds_bank_conflict_share")
            row["bound"] = "hbm" if row["hbm_traffic_frac"] >= 0.5 * row["frac"] and row["hbm_traffic_frac"] >= 0.3 else \
                "issue + lds + latency (operands cache-resident; results stream to HBM)"
            row["traffic_source"] = tr.get("source", "profiles/realdata_traffic.json") + " (recorded earlier, not re-measured in this run)"
        # the same batch with TWO calls in flight (rhip_pairwise_begin / _end): the host half of call i+1 overlaps
        # the kernels of call i; per-call period over 40 calls, result of the last one checked
        n_pipe, n_warm = max(4, 2 * D.reps), 8
        slots, prev = [res[0], None], None
        res[0] = None
        for it in range(-n_warm, n_pipe):  # warm-up: every slot's pinned staging exists before the clock starts
            if it == 0:
                D.barrier()
                t0 = time.perf_counter()
            cur = eng.pairwise_list_begin(op, plist, reuse=slots[it & 1])
            slots[it & 1] = None
            if prev is not None:
                slots[(it - 1) & 1] = prev.end()
            prev = cur
        slots[(n_pipe - 1) & 1] = prev.end()
        D.barrier()
        tp = (time.perf_counter() - t0) / n_pipe
        row["ms_batch_pipelined2"] = tp * 1e3
        row["ops_per_s_pipelined2"] = L.size / tp
        row["pipelined2_checksum_ok"] = bool(D.sum(float(slots[(n_pipe - 1) & 1].cardinalities().sum())) == csum)
        if hs is not None:
            row["cpu1_ops_per_s"] = cpu_pairs_rate(chk, hs, L, R, op, is64=is64)
            row["cpu_kind"] = chk.name
        out[f"{tag}_{op}"] = row
    # the ops of this configuration over the pair list in ONE batch (rhip_pairwise_multi): planned once
    if len(ops) > 1:
        res = [None]

        def mcall():
            res[0] = eng.pairwise_list(list(ops), plist, reuse=res[0])
        tmin, tmed = timed_calls(D, mcall)
        st = eng.last_stats()
        alg = D.sum(float(st["bytes_in"] + st["bytes_out"]))
        csum = D.sum(float(res[0].cardinalities().sum()))
        want = sum(out[f"{tag}_{op}"]["checksum"] for op in ops)
        sep = sum(out[f"{tag}_{op}"]["ms_batch_median"] for op in ops)
        if D.rank == 0:
            assert int(csum) == want, f"{name} multi: checksum {int(csum)} != sum of the single-op checksums {want}"
        out[f"{tag}_multi{len(ops)}"] = {"ops": list(ops), "pairs": int(L.size), "set_ops": int(L.size) * len(ops),
                                        "ms_batch_median": tmed * 1e3, "ms_batch_min": tmin * 1e3,
                                        "ms_sum_of_single_op_batches": sep, "ops_per_s": L.size * len(ops) / tmed,
                                        "alg_GBps": alg / tmed / 1e9, "frac": alg / tmed / 1e9 / HBM_PEAK_GBS,
                                        "checksum": int(csum), "checksum_ok": int(csum) == want}
        res[0] = None
    tmin, tmed = timed_calls(D, lambda: eng.pairwise_list_cardinality("and", plist))
    out[f"{tag}_and_cardinality"] = {"pairs": int(L.size), "ops_per_s": L.size / tmed, "ms_batch_median": tmed * 1e3,
                                     "ms_batch_min": tmin * 1e3}
    # The reference benchmark's own loop shape (benchmarks/benchmark.cpp:2035-2091 successive_and / successive_or): the
    # n - 1 ADJACENT pairs, each result materialised, its cardinality taken, freed.  Here: one batch over
    # rhip_pairlist_successive + the cardinalities read back.  199 pairs are almost pure fixed cost of a call -- this is
    # the row where one CPU core is expected to be at least as fast, reported as it comes (rank 0 only: nothing to shard).
    if not is64 and D.rank == 0:
        slist = eng.pairlist_successive(pool)
        n_succ = len(bufs) - 1
        sl, sr = np.arange(n_succ, dtype=np.uint32), np.arange(1, n_succ + 1, dtype=np.uint32)
        for op in ("and", "or"):
            res = [None]
            cards = [None]

            def scall():
                res[0] = eng.pairwise_list(op, slist, reuse=res[0])
                cards[0] = res[0].cardinalities()
            fn = scall
            fn(); fn()
            ts = []
            for _ in range(max(D.reps, 3)):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            tmed = float(np.median(ts))
            row = {"pairs": n_succ, "ms_batch_median": tmed * 1e3, "ms_batch_min": float(np.min(ts)) * 1e3,
                   "us_per_op": tmed * 1e6 / n_succ, "ops_per_s": n_succ / tmed, "checksum": int(cards[0].sum())}
            if hs is not None:
                rate = cpu_pairs_rate(chk, hs, sl, sr, op, budget=0.2)
                row["cpu1_ops_per_s"] = rate
                row["cpu1_us_per_op"] = 1e6 / rate
                row["gpu_over_cpu1"] = (n_succ / tmed) / rate
                want = sum(chk.cardinality(r) for r in (chk.op(op, hs[i], hs[i + 1]) for i in range(n_succ)))
                row["checksum_ok"] = bool(want == row["checksum"])
                assert row["checksum_ok"], f"{name} successive {op}: checksum differs from the reference"
            res[0] = None
            out[f"{tag}_successive_{op}"] = row
        slist.free()
    if is64:
        # BASELINE configs[4]: the many-way aggregation of the roaring64 bitmaps.  world > 1: bitmaps b mod world on rank
        # b mod world, the 48-bit-key chunks to their owners through the sparse exchange (RCCL), owners finalize; the
        # shares' cardinalities add up to the reference's fold (cpp/roaring/roaring64map.hh:1549-1670).
        from croaring_amd.distributed import many_sharded, shard_ids
        share = [None]

        def ustep():
            if D.world > 1:
                share[0] = many_sharded(eng, pool, "or", ids=shard_ids(len(bufs), D.rank, D.world))
            else:
                share[0] = eng.or_many(pool)
        tmin, tmed = timed_calls(D, ustep)
        card = int(D.sum(float(share[0].cardinalities()[0])))
        row = {"n": len(bufs), "ms_median": tmed * 1e3, "ms_min": tmin * 1e3, "cardinality": card,
               "parallelism": f"bitmaps b mod {D.world}, 48-bit-key chunks to key owners (sparse exchange over RCCL)" if D.world > 1 else "single GPU"}
        gp = os.path.join(ROOT, "tests", "golden", "c5_wikileaks64_pairs.npz")
        if os.path.exists(gp):
            row["cardinality_ok"] = bool(int(np.load(gp)["fold_or_card"][0]) == card)
            if D.rank == 0:
                assert row["cardinality_ok"], "C5 200-way union: cardinality differs from the reference's fold"
        if hs is not None:
            t0 = time.perf_counter()
            r = chk.or_many64(hs)
            row["cpu1_ms_fold"] = (time.perf_counter() - t0) * 1e3
            row["cardinality_ok"] = bool(row.get("cardinality_ok", True) and chk.cardinality64(r) == card)
            chk.free64(r)
        out[f"{tag}_union_{len(bufs)}"] = row
    if hs is not None:
        for h in hs:
            (chk.free64 if is64 else chk.free)(h)
    return out


def run_dropin_percall(chk, name: str, n_calls: int = 200):
    """Per-call latency of the CRoaring-named drop-in (include/roaring_hip_compat.h): ONE roaring_bitmap_and / _or on two
    host-resident reference structs = upload of both operands, a batch of one, download, a reference-layout result.  The
    CPU op it replaces beside it.  This is the compatibility path, not the product: the row exists so that the
    loss is a number.  Needs the real reference (its structs are the operands)."""
    import ctypes as C
    import croaring_amd
    from util import load_bundle
    if getattr(chk, "name", "") != "reference":
        return None
    lib = croaring_amd.load()
    bufs = load_bundle(name)
    hs = [chk.deserialize(b) for b in bufs]
    n = min(n_calls, len(hs) - 1)
    row = {"calls": n, "operands": name}
    for op in ("and", "or"):
        f = getattr(lib, f"roaring_bitmap_{op}")
        f.restype = C.c_void_p
        f.argtypes = [C.c_void_p, C.c_void_p]
        for i in range(3):  # warm-up: the lane's context, pinned staging and pools exist
            chk.free(f(hs[i], hs[i + 1]))
        t0 = time.perf_counter()
        for i in range(n):
            r = f(hs[i], hs[i + 1])
            chk.free(r)
        t_gpu = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for i in range(n):
            r = chk.op(op, hs[i], hs[i + 1])
            chk.free(r)
        t_cpu = (time.perf_counter() - t0) / n
        row[op] = {"dropin_us_per_call": t_gpu * 1e6, "croaring_us_per_call": t_cpu * 1e6, "dropin_over_cpu": t_gpu / t_cpu}
    for h in hs:
        chk.free(h)
    return row


def _time_steps(D: Dist, step, steps: int, warmup: int):
    for _ in range(warmup):
        step()
    ts = []
    for _ in range(steps):
        D.barrier()
        t0 = time.perf_counter()
        step()
        D.barrier()
        ts.append(D.max(time.perf_counter() - t0))
    return float(np.median(ts)), float(np.min(ts))


def gather_floats(D: Dist, x: float):
    """x of every rank, in rank order (world 1: [x])."""
    if D.world == 1:
        return [float(x)]
    t = D.torch.tensor([x], dtype=D.torch.float64, device=D.rdev)
    out = [D.torch.zeros_like(t) for _ in range(D.world)]
    D.dist.all_gather(out, t)
    return [float(v.item()) for v in out]


def measure_dense_a2a(eng, D: Dist, key_space: int = 4096, reps: int = 7):
    """The exchange step of the dense sharded or_many ALONE: one fixed-shape all_to_all_single of the [world x ceil(K /
    world), 1024] u64 table (32 MiB at K = 4096) on the engine's stream, max over ranks, median of `reps` -- the t_a2a of
    DESIGN 7a's model, which a single GPU cannot show.  None at world 1."""
    if D.world == 1:
        return None
    import torch
    from croaring_amd.distributed import dense_block
    B = dense_block(key_space, D.world)
    dev = eng.torch_device() if D.rdev == "cuda" else "cpu"
    with eng.torch_stream():
        src = torch.zeros((D.world * B, 1024), dtype=torch.int64, device=dev)
        dst = torch.empty_like(src)
        ts = []
        for it in range(reps + 2):
            D.barrier()
            t0 = time.perf_counter()
            D.dist.all_to_all_single(dst, src)
            torch.cuda.synchronize()
            dt = D.max(time.perf_counter() - t0)
            if it >= 2:
                ts.append(dt)
    return {"ms_median": float(np.median(ts)) * 1e3, "ms_min": float(np.min(ts)) * 1e3, "bytes_per_rank": int(src.numel() * 8),
            "bytes_to_peers": int(src.numel() * 8 * (D.world - 1) // D.world)}


def run_ormany(args, eng, D: Dist, steps: int, warmup: int, chk=None):
    """C4, strong scaling: --bitmaps sparse bitmaps in total, rank r holds bitmaps b with b mod world == r; one step
    = one or_many over ALL of them.  world > 1: per-rank partial chunks written into the dense send table -> ONE
    fixed-shape all-to-all over RCCL -> owner finalize, one host wait at the end (croaring_amd.distributed).
    world == 1: rhip_or_many, and beside it the SAME sharded pipeline on a one-rank group -- with the (identity)
    collective skipped and with it issued on a 1-rank nccl group -- so that the pipeline's fixed cost is a number."""
    import croaring_amd
    from croaring_amd.distributed import many_sharded
    n_local = (args.bitmaps - D.rank + D.world - 1) // D.world
    blob, offs = croaring_amd.synth_sparse_portable(D.rank, D.world, n_local)
    pool = eng.pool_from_blob(blob, offs)
    payload = pool.payload_bytes()
    out = [None]

    def step():
        out[0] = many_sharded(eng, pool, "or", key_space=4096) if D.world > 1 else eng.or_many(pool)

    tmed, tmin = _time_steps(D, step, steps, warmup)
    tot_payload = D.sum(float(payload))
    card = D.sum(float(out[0].cardinalities()[0]))
    bytes_out = D.sum(float(eng.last_stats()["bytes_out"]))
    row = {"bitmaps": args.bitmaps, "containers": 32 * args.bitmaps, "ms_median": tmed * 1e3, "ms_min": tmin * 1e3,
           "ops_per_s": 1 / tmed, "alg_GBps": (tot_payload + bytes_out) / tmed / 1e9,
           "result_cardinality": int(card), "scaling": "strong",
           "pool_layout": {"payload_align": int(pool.payload_align), "arena_over_payload": round(pool.arena_bytes() / max(1, payload), 4)},
           "parallelism": f"bitmaps b mod {D.world}; dense key-owner all-to-all over RCCL, one host wait" if D.world > 1 else "single GPU"}
    row["frac"] = row["alg_GBps"] / HBM_PEAK_GBS
    if D.world > 1:
        # DESIGN 7a beside the measurement: T(N) = 0.08 + 0.52 / N + t_a2a(N) + 0.03 ms (fixed part of stage 1, the rank's share
        # of the reduction, the exchange, stage 3) with t_a2a MEASURED here on its own
        a2a = measure_dense_a2a(eng, D)
        row["t_a2a"] = a2a
        row["model_ms"] = {"N": D.world, "stage1_fixed": 0.08, "stage1_share": 0.52 / D.world, "t_a2a_measured": a2a["ms_median"],
                           "stage3": 0.03, "predicted": 0.08 + 0.52 / D.world + a2a["ms_median"] + 0.03, "measured": tmed * 1e3}
    gp = os.path.join(ROOT, "tests", "golden", "c4_or_many.npz")
    if os.path.exists(gp) and args.bitmaps == 100000:
        row["cardinality_ok"] = bool(int(np.load(gp)["or_many_100000"][0]) == int(card))
        assert row["cardinality_ok"], "C4 or_many cardinality differs from the reference fixture"
    if D.world == 1:
        # the sharded pipeline at world 1: what a rank pays on top of its share of the reduction
        dist, made = D.dist, False
        try:
            if not dist.is_initialized():
                import socket
                s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                        device_id=D.torch.device("cuda", D.torch.cuda.current_device()))
                made = True
            for tag, force in (("sharded_w1", False), ("sharded_w1_nccl", True)):
                def sstep():
                    out[0] = many_sharded(eng, pool, "or", key_space=4096, force_collective=force)
                m, mn = _time_steps(D, sstep, steps, warmup)
                ok = int(out[0].cardinalities()[0]) == int(card)
                row[tag] = {"ms_median": m * 1e3, "ms_min": mn * 1e3, "vs_or_many": m / tmed, "cardinality_ok": bool(ok)}
                assert ok, "sharded or_many at world 1 differs from or_many"
        except Exception as e:  # (no RCCL in this process: the number is reported as missing, the headline is unaffected)
            row["sharded_w1_error"] = str(e)[:120]
        finally:
            if made:
                dist.destroy_process_group()
    if chk is not None and D.rank == 0 and D.world == 1:
        # CPU reference on a bounded sample: the first 10 000 bitmaps (deserialisation untimed)
        n = min(10000, n_local)
        mv = memoryview(blob)
        hs = [chk.deserialize(bytes(mv[int(offs[b]):int(offs[b + 1])])) for b in range(n)]
        t0 = time.perf_counter()
        r = chk.or_many(hs)
        row["cpu1_ms_first_10000"] = (time.perf_counter() - t0) * 1e3
        row["cpu_kind"] = chk.name
        chk.free(r)
        for h in hs:
            chk.free(h)
    return row


def sparse_pool(eng, first: int, stride: int, count: int, chunk: int = 100000):
    """A pool of `count` bitmaps first, first + stride, ... of the C4 generator, built 100 000 bitmaps at a time (1.7 GB
    of portable images per piece on the host) and merged on the device (rhip_pool_select)."""
    import croaring_amd
    parts = []
    for c0 in range(0, count, chunk):
        n = min(chunk, count - c0)
        blob, offs = croaring_amd.synth_sparse_portable(first + stride * c0, stride, n)
        parts.append(eng.pool_from_blob(blob, offs))
        del blob
    if len(parts) == 1:
        return parts[0]
    src_pool = np.concatenate([np.full(len(p), i, np.uint32) for i, p in enumerate(parts)])
    src_bm = np.concatenate([np.arange(len(p), dtype=np.uint32) for p in parts])
    merged = eng.pool_select(parts, src_pool, src_bm)
    for p in parts:
        p.free()
    return merged


def run_ormany_x10(args, eng, D: Dist):
    """The C4 generator at 10^6 bitmaps (32 M array containers, 16.4 GB of payload): the many-way row where the
    reduction is big enough for N GPUs to split it -- DESIGN 7a's model gives 6.9 ms / N + 0.2 ms against C4's
    0.69 / N + 0.13.  Same pipeline as c4_or_many; the cardinality is checked against the reference's answer
    (tests/golden/c4x10_or_many.npz, oracle/gen_golden.py c4x10)."""
    from croaring_amd.distributed import many_sharded
    n_total = 10 * args.bitmaps
    n_local = (n_total - D.rank + D.world - 1) // D.world
    t0 = time.perf_counter()
    pool = sparse_pool(eng, D.rank, D.world, n_local)
    t_build = time.perf_counter() - t0
    payload = pool.payload_bytes()
    out = [None]

    def step():
        out[0] = many_sharded(eng, pool, "or", key_space=4096) if D.world > 1 else eng.or_many(pool)
    tmed, tmin = _time_steps(D, step, 5, 1)
    card = int(D.sum(float(out[0].cardinalities()[0])))
    tot_payload = D.sum(float(payload))
    row = {"bitmaps": n_total, "ms_median": tmed * 1e3, "ms_min": tmin * 1e3, "alg_GBps": tot_payload / tmed / 1e9,
           "frac": tot_payload / tmed / 1e9 / HBM_PEAK_GBS, "result_cardinality": card, "scaling": "strong",
           "build_s_untimed": t_build,
           "model_ms": {"N": D.world, "predicted": 6.9 / D.world + (0.2 if D.world > 1 else 0.0), "at_8": 6.9 / 8 + 0.2}}
    gp = os.path.join(ROOT, "tests", "golden", "c4x10_or_many.npz")
    if os.path.exists(gp) and n_total == 1000000:
        row["cardinality_ok"] = bool(int(np.load(gp)["or_many"][0]) == card)
        if D.rank == 0:
            assert row["cardinality_ok"], "C4 x 10 or_many cardinality differs from the reference fixture"
    out[0] = None
    pool.free()
    return row


def run_shard_stages(args, eng, D: Dist, n_bitmaps: int):
    """What ONE rank of an N-rank group would pay for the sharded or_many, measured on one GPU: stage 1
    (rhip_many_partials_dense over rank 0's pool -- the bitmaps b mod N == 0 -- into a world = N send table) and stage 3
    (rhip_many_finalize_dense over a world = N receive table), each timed to completion.  The all-to-all between them
    is what a single GPU cannot show.  Beside each N: DESIGN 7a's model for the same two stages."""
    import torch
    import croaring_amd
    from croaring_amd.distributed import dense_block, shard_ids
    dev = eng.torch_device()
    rows = {}
    for N in (1, 2, 4, 8):
        # rank 0's own pool, as bench.py --gpus N builds it: bitmaps 0, N, 2 N, ... (no selection list: the whole pool)
        blob, offs = croaring_amd.synth_sparse_portable(0, N, (n_bitmaps + N - 1) // N)
        pool = eng.pool_from_blob(blob, offs)
        del blob
        B = dense_block(4096, N)
        with eng.torch_stream():
            table = torch.empty((N * B, 1024), dtype=torch.int64, device=dev)
        t1, t3 = [], []
        for it in range(9):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.many_partials_dense("or", pool, None, 4096, N, table.data_ptr())
            eng.synchronize()
            ta = time.perf_counter()
            res = eng.many_finalize_dense("or", False, N, 0, B, table.data_ptr())  # (rows of N "sources": this rank's own, N times over)
            tb = time.perf_counter()
            res.free()
            if it >= 3:
                t1.append(ta - t0)
                t3.append(tb - ta)
        rows[str(N)] = {"stage1_ms": float(np.median(t1)) * 1e3, "stage3_ms": float(np.median(t3)) * 1e3,
                        "model_stage1_ms": 0.10 + 0.69 / N, "model_stage3_ms": 0.03}
        pool.free()
    return rows


# ----------------------------------------------------------------------------- main
def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner when a
    communicator is created), so file descriptor 1 is pointed at stderr for the whole run and the line goes to a
    private duplicate of the original stdout."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def main():
    args = parse_args()
    if args.cpu_baseline_only:  # (no torch, no HIP, no NCCL in this process: only the CPU reference and its worker processes)
        print(json.dumps(cpu_baseline(args, args.cpu_seconds)), flush=True)
        return
    maybe_spawn(args)
    real_stdout = claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a mislabelled line")
    import torch  # first: the engine then binds to the HIP runtime torch already loaded
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    share = bool(args.ranks_share_device)
    if not share and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} device(s) visible "
                         "(--ranks-share-device runs the N-rank script on one GPU over gloo, as a dry run)")
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import croaring_amd
    if share:
        # the dry run's batches are a few hundred container pairs: the library would MERGE their class kernels into one
        # launch, and k_bb -- the kernel every rank reports -- would have no events of its own.  Dry run only.
        os.environ.setdefault("RHIP_MERGE_CLASSES", "0")
    eng = croaring_amd.Engine(dev_index)
    eng.set_timing(True)
    D = Dist(rank, world, torch, dist, reps=args.reps, host_tensors=share)
    transport = ("gloo-staged (dry run: %d ranks share one GPU, exchange through host memory -- NOT a scaling number)" % world) if share \
        else ("rccl" if world > 1 else "none (one rank)")
    chk = None
    if rank == 0 and not args.no_cpu:
        from oracle.pyoracle import best_checker
        chk = best_checker()

    if args.workload == "ormany":
        row = run_ormany(args, eng, D, args.steps, args.warmup, chk)
        out = {"metric": METRIC, "value": row["ops_per_s"], "unit": "or_many-ops/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": row["ms_median"], "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic", "transport": transport,
               "config": {"workload": f"C4 or_many over {args.bitmaps} sparse bitmaps x 32 array containers", **row}}
        if rank == 0:
            print(json.dumps(out), file=real_stdout, flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    pool = eng.pool_synth_bitset(args.pool, args.containers, SEED + 1000003 * rank)
    results = {"and": None, "or": None}
    placement = {}
    bb_ms, bb_pairs = [], []

    def step(i: int, timed: bool):
        # the 2 x rounds calls of a step are issued with two in flight (rhip_pairwise_begin / _end): the planning of
        # call k+1 runs under the bitset kernel of call k.  Every call of the step has ended when the step returns.
        pending = None

        def finish(p):
            op, b = p
            results[op] = b.end()
            if timed:
                st = eng.last_stats()
                bb_ms.append(st["ms_bitset_kernel"])
                bb_pairs.append(st["n_bitset_pairs"])

        for r in range(args.rounds):
            for j, op in enumerate(("and", "or")):
                lhs, rhs = schedule(((i * args.rounds + r) * 2 + j) * args.pairs, args.pairs, args.pool)
                fresh = results[op] is None
                b = eng.pairwise_begin(op, pool, lhs, pool, rhs, reuse=results[op])
                if fresh:  # the library has just placed this result pool's arena (warm-up, untimed): keep its probe rates
                    placement[op] = eng.last_placement()
                results[op] = None
                if pending is not None:
                    finish(pending)
                pending = (op, b)
        finish(pending)

    # (diagnostic, --arena-tries > 0: the CALLER picks the two result pools by measurement, Engine.pairwise_placed -- round 3's
    # start-up; by default the library places each result arena itself when it allocates it)
    arena_probe = []
    if args.arena_tries > 0:
        lhs, rhs = schedule(0, args.pairs, args.pool)
        (results["and"], results["or"]), arena_probe = eng.pairwise_placed("or", pool, lhs, pool, rhs, tries=args.arena_tries,
                                                                           keep=2, timing_after=True)
    fresh_ms = {}
    if results["and"] is None:
        # untimed start-up, whatever --warmup says: the two result pools that every later step recycles are allocated here --
        # one SYNCHRONOUS call each (the library places a new arena only when no batch of the context is in flight).
        # Allocation is not part of a step; what a caller WITHOUT `reuse` pays for it is reported: c2_fresh_result_pool_ms.
        lhs, rhs = schedule(0, args.pairs, args.pool)
        for op in ("and", "or"):
            t_ = time.perf_counter()
            results[op] = eng.pairwise(op, pool, lhs, pool, rhs)
            fresh_ms[op] = (time.perf_counter() - t_) * 1e3
            placement[op] = eng.last_placement()
        step(0, False)
    for i in range(args.warmup):
        step(i, False)
    D.barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i, True)
    D.barrier()
    dt = D.max(time.perf_counter() - t0)

    # sanity: every result container of the last OR batch is a bitset
    assert results["or"].type_counts() == (args.pairs * args.containers, 0, 0)
    results["and"].free()
    results["or"].free()
    # What a caller WITHOUT `reuse` pays per call in steady state -- allocate, compute, free, again: the arena of a freed
    # result pool is parked with the context (round 6) and taken back, as placed, by the next result pool for the same
    # operand: no allocation, no probe.  (untimed region of the bench: after the clock stopped)
    if fresh_ms:
        lhs, rhs = schedule(0, args.pairs, args.pool)
        for op in ("and", "or"):
            ts = []
            for _ in range(3):
                t_ = time.perf_counter()
                r_ = eng.pairwise(op, pool, lhs, pool, rhs)
                ts.append((time.perf_counter() - t_) * 1e3)
                r_.free()
            fresh_ms["steady_" + op] = float(np.median(ts))

    ops_per_step = 2 * args.pairs * args.rounds
    total_ops = ops_per_step * args.steps * world
    ms_kernel = float(np.mean(bb_ms)) if bb_ms else 0.0
    pairs_per_launch = float(np.mean(bb_pairs)) if bb_pairs else 0.0
    achieved = (pairs_per_launch * BB_BYTES_PER_PAIR) / (ms_kernel * 1e-3) / 1e9 if ms_kernel > 0 else 0.0
    per_rank_frac = gather_floats(D, achieved / HBM_PEAK_GBS)
    per_rank_ms = gather_floats(D, ms_kernel)
    traffic, traffic_source = None, None
    tp = os.path.join(ROOT, "profiles", "bb_traffic.json")
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            traffic = tj.get("hbm_bytes_per_launch")
            traffic_source = (f"profiles/bb_traffic.json <- {tj.get('source', 'profiles/r01_pmc_summary.md')} "
                              "(rocprofv3 --pmc passes recorded earlier; NOT re-measured in this run)")
        except Exception:
            traffic = None
    traffic_live = None
    if world == 1 and rank == 0 and not args.no_live_traffic:
        traffic_live, why = live_bb_traffic(args)
        if traffic_live is not None:
            stored = traffic
            traffic = traffic_live["hbm_bytes_per_launch"] * (pairs_per_launch / float(args.pairs * args.containers) if pairs_per_launch else 1.0)
            traffic_source = ("LIVE: two `rocprofv3 --pmc` child runs of this headline made by this bench.py run (FETCH_SIZE, WRITE_SIZE; "
                              "separate passes, --kernel-trace only; KiB units, FETCH_SIZE x 2 per the gfx950 correction; per 250-pair "
                              f"launch x the {pairs_per_launch / float(args.pairs * args.containers):.3g} launches' worth of pairs the timed k_bb "
                              f"launches held on average); calibration {traffic_live['calibration']}; stored pass for comparison: {stored}")
        elif traffic_source:
            traffic_source += f" [live measurement failed: {why}]"
    out = {
        "metric": METRIC,
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
        "transport": transport,
        "collective": None if world == 1 else {
            "backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_ranks": dist.get_world_size() if dist.get_backend() == "nccl" else 0,
            "note": "torch.distributed's nccl backend IS RCCL on ROCm; rccl_ranks = the size of the communicator the many-way rows exchange over "
                    "(0: the gloo dry run on one GPU)"},
        "per_rank": None if world == 1 else {
            "k_bb_frac": [round(v, 4) for v in per_rank_frac], "k_bb_avg_launch_ms": [round(v, 4) for v in per_rank_ms],
            "note": "every rank's own HIP-event average of k_bb over the timed region (roofline.* is rank 0's)"},
        "config": {"workload": f"C2 synthetic bitset-only: pool {args.pool} bitmaps x {args.containers} bitset "
                               f"containers (density 0.5), batched pairwise AND+OR, {args.pairs} pairs per call, "
                               f"{args.rounds} x (AND call + OR call) per step, two calls in flight "
                               f"(rhip_pairwise_begin / _end), every call ended inside its step",
                   "ops_per_step": ops_per_step,
                   "timed_region_s": dt,
                   "algorithmic_GBps": total_ops * args.containers * BB_BYTES_PER_PAIR / dt / 1e9,
                   "parallelism": f"pairs sharded over {world} GPU(s), no data-path collective",
                   "c2_fresh_result_pool_ms": {"note": "and / or: the FIRST 250-pair call, which has to allocate (and, >= 2 GiB, place by measurement) its "
                                                       "result pool; steady_and / steady_or: a caller that never passes `reuse` -- call, free, call "
                                                       "again -- in steady state (the freed pool's placed arena is parked with the context and taken back); "
                                                       f"a call with `reuse` is ms_per_step / {2 * args.rounds}", **{k: round(v, 2) for k, v in fresh_ms.items()}},
                   "result_arena_placement": {"by": "library (rhip place_arena, untimed warm-up)" if args.arena_tries == 0 else "caller (Engine.pairwise_placed)",
                                              "probe_GBps_of_each_candidate": placement if args.arena_tries == 0 else None,
                                              "k_bb_ms_of_each_try": arena_probe or None,
                                              "note": "the two recycled result pools: each arena is the fastest of the candidates the library "
                                                      "probed when it allocated it (the physical distance between operand and result stream "
                                                      "moves k_bb by up to 17 %, DESIGN 4a); RHIP_ARENA_TRIES=0 takes the first allocation"}},
        "roofline": {"bound": "hbm", "kernel": "k_bb (bitset x bitset fused op+popcount)", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_source, "avg_launch_ms": ms_kernel,
                     "pairs_per_launch": pairs_per_launch, "launches_timed": len(bb_ms)},
    }
    del pool
    detail = {}
    if not args.no_secondary:
        sec = {}
        eng.set_timing(False)  # HIP-event timing (extra records, blocking stream wait) is for the headline's roofline only
        sec.update(run_realdata(eng, D, "weather_sept_85", "c3", chk))
        sec.update(run_realdata(eng, D, "census1881", "c1", chk))
        sec.update(run_realdata(eng, D, "wikileaks-noquotes x10 (roaring64)", "c5", chk, is64=True, ops=("and", "or")))
        if rank == 0 and chk is not None:
            try:
                r = run_dropin_percall(chk, "census1881")
                if r:
                    sec["dropin_percall_us"] = r
            except Exception as e:  # (a compatibility row: its failure is reported, it does not take the line down)
                sec["dropin_percall_us"] = {"error": str(e)[:160]}
        sec["c4_or_many"] = run_ormany(args, eng, D, steps=max(3, min(10, args.reps)), warmup=2, chk=chk)
        # (the two rows below are evidence for the scaling model, not the headline: a failure in them is reported as a
        # row, it does not take the line down -- except a WRONG cardinality, which is asserted inside)
        if not args.no_x10:
            try:
                sec["c4x10_or_many"] = run_ormany_x10(args, eng, D)
            except AssertionError:
                raise
            except Exception as e:
                sec["c4x10_or_many"] = {"error": str(e)[:200]}
        if world == 1:
            try:
                sec["c4_shard_stages"] = run_shard_stages(args, eng, D, args.bitmaps)
            except Exception as e:
                sec["c4_shard_stages_error"] = {"error": str(e)[:200]}
        detail = sec
        # The line itself carries one compact row per configuration -- [ms per batch (median, whole call), fraction of the
        # HBM peak (algorithmic bytes / time / 8 TB/s), checksum ok] -- so that it stays well under the 8 KB of stdout the
        # driver keeps; every other figure of a row goes to stderr (`BENCH_DETAIL {...}`) and to gpurun_out/bench_detail.json.
        summ = {}
        for k, r in sec.items():
            if not isinstance(r, dict):
                continue
            if "error" in r:
                summ[k] = r["error"][:80]
                continue
            if k == "dropin_percall_us":
                summ[k] = {o: [round(r[o]["dropin_us_per_call"], 1), round(r[o]["croaring_us_per_call"], 2)] for o in ("and", "or") if o in r}
                continue
            if k == "c4_shard_stages":
                summ[k] = {n: [round(v["stage1_ms"], 3), round(v["stage3_ms"], 3), round(v["model_stage1_ms"] + v["model_stage3_ms"], 3)]
                           for n, v in r.items()}
                continue
            ms = r.get("ms_batch_median", r.get("ms_median"))
            ok = r.get("checksum_ok", r.get("cardinality_ok"))
            row = [None if ms is None else round(ms, 4), None if r.get("frac") is None else round(r["frac"], 4), ok]
            if "ms_adhoc_list" in r:
                row += [round(r["ms_adhoc_list"], 4), round(r.get("ms_batch_pipelined2", 0.0), 4),
                        None if "hbm_traffic_frac" not in r else round(r["hbm_traffic_frac"], 3),
                        None if "ms_first_call" not in r else round(r["ms_first_call"], 4)]
            if "sharded_w1" in r:
                row += [round(r["sharded_w1"]["vs_or_many"], 3)]
            if r.get("t_a2a"):  # N > 1: the exchange measured alone, and DESIGN 7a's model with it
                row += [round(r["t_a2a"]["ms_median"], 4), round(r["model_ms"]["predicted"], 4)]
            if "us_per_op" in r:
                row += [round(r["us_per_op"], 3), None if "cpu1_us_per_op" not in r else round(r["cpu1_us_per_op"], 3)]
            summ[k] = row
        out["config"]["secondary_summary"] = {
            "rows": summ,
            "columns": "ms per batch (median, whole call incl. the wait; realdata: a REPEATED batch over a prepared pair list, whose plan "
                       "is kept with the list -- ms_planned) | fraction of the 8 TB/s HBM peak | checksum / "
                       "cardinality equal to the reference's | realdata only: ms with the pair list handed over per call instead of "
                       "prepared once (plans every time), ms per call with two calls in flight, HBM TRAFFIC (stored PMC pass) / time / 8 TB/s -- the "
                       "second column counts algorithmic bytes, which on these cache-resident sets is not HBM utilisation --, ms of the "
                       "FIRST call over the prepared list (planning kernels included: ms_first_call) | c4: sharded pipeline at world 1 / or_many; "
                       "N > 1: ms of the dense all-to-all measured alone (t_a2a), ms DESIGN 7a's model predicts with it | "
                       "c4_shard_stages: N -> [stage 1 ms, stage 3 ms, DESIGN 7a model ms] for one rank of N on this GPU | "
                       "*_successive_*: the n - 1 adjacent pairs as one batch + cardinalities read back (the reference benchmark's "
                       "successive_and / _or loop), then us per op here, us per op of CRoaring on one core | dropin_percall_us: op -> "
                       "[us per call through the CRoaring-named per-call drop-in, us per call of CRoaring]",
            "note": "realdata: ALL unordered pairs in one batched call per op over a prepared pair list; pairs partitioned over ranks"}
    # cpu_baseline: rank 0's host cores, whatever the world size (the other ranks wait at the barrier below)
    if rank == 0 and not args.no_cpu:
        if world == 1:
            cb = cpu_baseline(args, args.cpu_seconds)
        else:
            # N > 1: in a fresh process -- the sweep forks worker processes, and forking a process that holds an RCCL
            # communicator is not something to find out about inside the scaling run; bounded, and a failure is a row
            import subprocess
            try:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-seconds", str(args.cpu_seconds),
                                    "--containers", str(args.containers), "--pool", str(args.pool), "--pairs", str(args.pairs)],
                                   capture_output=True, text=True, timeout=180,
                                   env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")})
                js = [l for l in p.stdout.split("\n") if l.startswith("{")]
                if p.returncode != 0 or not js:
                    raise RuntimeError(f"--cpu-baseline-only exited {p.returncode}: {p.stderr[-400:]!r}")
                cb = json.loads(js[-1])
            except Exception as e:
                cb = None
                out["cpu_baseline"] = {"error": str(e)[:600]}
    if rank == 0 and not args.no_cpu and cb is not None:
        detail["cpu_baseline_full"] = cb
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "host_threads")}
        out["cpu_baseline"]["one_core_ops_per_s"] = round(cb["one_core"]["ops_per_s_median"], 1)
        out["cpu_baseline"]["sweep_ops_per_s"] = {T: round(r["ops_per_s_median"]) for T, r in cb["sweep"].items()}
        out["cpu_baseline"]["isa_1core_ops_per_s"] = {k: round(v.get("ops_per_s_median_1core", 0.0), 1) for k, v in (cb["isa_1core"] or {}).items()}
        out["cpu_baseline"]["note"] = ("CRoaring is malloc-bound here: every op allocates and writes a 32 MiB result, so worker processes "
                                       "stop scaling at ~16 on this host (page faults, not cores)")
        out["cpu_baseline"]["sample"] = cb["sample"][:300]
    elif rank == 0 and "cpu_baseline" not in out:
        out["cpu_baseline"] = None
    D.barrier()
    if rank == 0:
        line = json.dumps(out)
        try:
            print("BENCH_DETAIL " + json.dumps(detail), file=sys.stderr, flush=True)
            dd = os.path.join(ROOT, "gpurun_out")
            if os.path.isdir(dd):
                with open(os.path.join(dd, "bench_detail.json"), "w") as f:
                    json.dump({"line": out, "detail": detail}, f)
        except Exception:
            pass
        print(line, file=real_stdout, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
