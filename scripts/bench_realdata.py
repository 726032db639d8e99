"""Secondary measurements (not the bench.py line): SURVEY §8d C1/C3 realdata all-pairs, C4 sparse or_many,
next to the CPU reference (oracle/_ref when present).  Prints one JSON object per line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import croaring_amd
from util import load_bundle, all_pairs, OPS
from oracle.pyoracle import best_checker

eng = croaring_amd.Engine(0)
eng.set_timing(True)
chk = best_checker()
REPS = 5


def timed(fn, reps=REPS):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts), float(np.median(ts))


def cpu_rate(hs, lhs, rhs, op, budget=2.0):
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget:
        for i, j in zip(lhs, rhs):
            r = chk.op(op, hs[i], hs[j]); chk.cardinality(r); chk.free(r); n += 1
            if n % 256 == 0 and time.perf_counter() - t0 > budget: break
    return n / (time.perf_counter() - t0)


for name in sys.argv[1:] or ["census1881", "weather_sept_85", "wikileaks-noquotes", "census-income"]:
    if name.startswith("c4") or name == "c5":
        continue
    bufs = load_bundle(name)
    pool = eng.pool_from_serialized(bufs)
    lhs, rhs = all_pairs(len(bufs))
    hs = [chk.deserialize(b) for b in bufs]
    for op in OPS:
        res = [None]
        def run():
            res[0] = eng.pairwise(op, pool, lhs, pool, rhs, reuse=res[0])
        tmin, tmed = timed(run)
        st = eng.last_stats()
        alg = st["bytes_in"] + st["bytes_out"]
        out = {"dataset": name, "op": op, "pairs": len(lhs), "gpu_ops_per_s": len(lhs) / tmin, "gpu_ms_batch": tmin * 1e3,
               "algorithmic_bytes": alg, "gpu_GBps": alg / tmin / 1e9, "matched_pairs": st["matched_pairs"],
               "bitset_pairs": st["n_bitset_pairs"], "passthrough": st["passthrough"],
               "cpu1_ops_per_s": cpu_rate(hs, lhs, rhs, op), "cpu_kind": chk.name}
        print(json.dumps(out), flush=True)
    tmin, _ = timed(lambda: eng.pairwise_cardinality("and", pool, lhs, pool, rhs))
    print(json.dumps({"dataset": name, "op": "and_cardinality", "pairs": len(lhs), "gpu_ops_per_s": len(lhs) / tmin,
                      "gpu_ms_batch": tmin * 1e3}), flush=True)
    for nm, fn, cf in (("or_many", eng.or_many, chk.or_many), ("xor_many", eng.xor_many, chk.xor_many)):
        tmin, _ = timed(lambda: fn(pool))
        t0 = time.perf_counter(); r = cf(hs); tc = time.perf_counter() - t0; chk.free(r)
        print(json.dumps({"dataset": name, "op": nm, "n": len(bufs), "gpu_ms": tmin * 1e3, "cpu1_ms": tc * 1e3}), flush=True)
    for h in hs: chk.free(h)

if any(a == "c5" for a in sys.argv[1:]) or len(sys.argv) == 1:
    # C5: 64-bit bitmaps = wikileaks-noquotes replicated into 10 high-32 buckets (v + (r << 32))
    import struct
    base = load_bundle("wikileaks-noquotes")
    bufs64 = [struct.pack("<Q", 10) + b"".join(struct.pack("<I", r) + b for r in range(10)) for b in base]
    pool = eng.pool_from_serialized64(bufs64)
    lhs, rhs = all_pairs(len(bufs64))
    hs = [chk.deserialize64(b) for b in bufs64]
    for op in ("and", "or"):
        res = [None]
        def run():
            res[0] = eng.pairwise(op, pool, lhs, pool, rhs, reuse=res[0])
        tmin, _ = timed(run)
        st = eng.last_stats()
        n, t0 = 0, time.perf_counter()
        for i, j in zip(lhs[:4000], rhs[:4000]):
            r = chk.op64(op, hs[i], hs[j]); chk.cardinality64(r); chk.free64(r); n += 1
        cpu = n / (time.perf_counter() - t0)
        k = 777
        ok = res[0].serialize(k) == chk.serialize64(chk.op64(op, hs[lhs[k]], hs[rhs[k]]))
        print(json.dumps({"dataset": "C5 roaring64 wikileaks-noquotes x10 buckets", "op": op, "pairs": len(lhs),
                          "gpu_ops_per_s": len(lhs) / tmin, "gpu_ms_batch": tmin * 1e3,
                          "gpu_GBps": (st["bytes_in"] + st["bytes_out"]) / tmin / 1e9, "matched_pairs": st["matched_pairs"],
                          "cpu1_ops_per_s": cpu, "cpu_kind": chk.name, "sample_equal": bool(ok)}), flush=True)
    tmin, _ = timed(lambda: eng.or_many(pool))
    t0 = time.perf_counter(); r = chk.or_many64(hs); tc = time.perf_counter() - t0
    print(json.dumps({"dataset": "C5 roaring64 wikileaks-noquotes x10 buckets", "op": "or_many (200-way)",
                      "gpu_ms": tmin * 1e3, "cpu1_ms_fold": tc * 1e3,
                      "equal_card": bool(eng.or_many(pool).cardinalities()[0] == chk.cardinality64(r))}), flush=True)

if any(a.startswith("c4") for a in sys.argv[1:]) or len(sys.argv) == 1:
    arg = [a for a in sys.argv[1:] if a.startswith("c4")]
    NB = int(arg[0].split("=")[1]) if arg and "=" in arg[0] else 100000
    # C4: NB sparse bitmaps, 32 array containers each (keys stratified over [0,4096), card uniform
    # in [1,512], values stratified over [0,65536)) -- ~NB*32 containers, array-dominant.
    t0 = time.perf_counter()
    rng = np.random.default_rng(4)
    NK = 32
    keys = (np.arange(NK, dtype=np.uint32)[None, :] * 128 + rng.integers(0, 128, (NB, NK), dtype=np.uint32)).astype(np.uint16)
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
    h32[:, 0] = 12346; h32[:, 1] = NK
    h32[:, 2:2 + NK] = keys.astype(np.uint32) | ((cards - 1).astype(np.uint32) << 16)
    inner = np.concatenate([np.zeros((NB, 1), np.int64), np.cumsum(cards.astype(np.int64) * 2, 1)[:, :-1]], 1) + hdr
    h32[:, 2 + NK:] = inner.astype(np.uint32)
    hb = h32.view(np.uint8)
    idx = (offs[:-1, None] + np.arange(hdr)[None, :]).ravel()
    blob[idx] = hb.ravel()
    # payload bytes: values of bitmap b start at offs[b]+hdr, contiguous
    bm_of_val = cid // NK
    vstart = np.concatenate([[0], np.cumsum(cards.sum(1).astype(np.int64))])
    pos = offs[bm_of_val] + hdr + 2 * (np.arange(total, dtype=np.int64) - vstart[bm_of_val])
    blob[pos] = (vals & 0xFF).astype(np.uint8); blob[pos + 1] = (vals >> 8).astype(np.uint8)
    tgen = time.perf_counter() - t0
    t0 = time.perf_counter()
    pool = eng.pool_from_packed(blob, offs[:-1], per_bm)
    tup = time.perf_counter() - t0
    tmin, tmed = timed(lambda: eng.or_many(pool), reps=3)
    res = eng.or_many(pool)
    st = eng.last_stats()
    payload = pool.payload_bytes()
    out = {"dataset": f"C4 synthetic sparse {NB} bitmaps x {NK} array containers", "op": "or_many", "containers": int(NB * NK),
           "payload_bytes": payload, "gpu_ms": tmin * 1e3, "gpu_GBps": (payload + st["bytes_out"]) / tmin / 1e9,
           "result_containers": st["result_containers"], "result_card": int(res.cardinalities()[0]),
           "gen_s": tgen, "upload_s": tup}
    if os.environ.get("SKIP_CPU"):
        print(json.dumps(out), flush=True)
        sys.exit(0)
    # CPU reference on the same bytes
    t0 = time.perf_counter()
    mv = memoryview(blob)
    hs = [chk.deserialize(bytes(mv[int(offs[b]):int(offs[b + 1])])) for b in range(NB)]
    t1 = time.perf_counter(); r = chk.or_many(hs); tc = time.perf_counter() - t1
    out.update(cpu1_ms=tc * 1e3, cpu_kind=chk.name, cpu_card=int(chk.cardinality(r)),
               equal=bool(res.serialize(0) == chk.serialize(r)))
    print(json.dumps(out), flush=True)
