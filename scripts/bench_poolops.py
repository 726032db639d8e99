"""Secondary measurements of the reshaping / (de)serialization entry points on one MI355X (JSON lines).
    python scripts/bench_poolops.py > gpurun_out/poolops.jsonl
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def best(fn, reps=5):
    ts = []
    out = None
    for _ in range(reps):
        t = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t)
    return min(ts), out


def main():
    import torch  # noqa: F401
    import croaring_amd
    from util import all_pairs, load_bundle
    eng = croaring_amd.Engine()
    for name in ("weather_sept_85", "census1881"):
        bufs = load_bundle(name)
        nbytes = sum(len(b) for b in bufs)
        lens = np.array([len(b) for b in bufs], dtype=np.uint64)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        blob = np.frombuffer(b"".join(bufs), dtype=np.uint8)
        t_host, H = best(lambda: eng.pool_from_serialized(bufs))
        t_dev, P = best(lambda: eng.pool_from_blob(blob, offs))
        assert P.serialize_all() == bufs
        print(json.dumps({"what": "load", "dataset": name, "bytes": nbytes, "host_parser_ms": t_host * 1e3,
                          "device_parser_ms": t_dev * 1e3, "device_GBps": nbytes / t_dev / 1e9}), flush=True)
        lhs, rhs = all_pairs(len(bufs))
        R = eng.pairwise("or", H, lhs, H, rhs)
        t_one, _ = best(lambda: [R.serialize(k) for k in range(0, len(lhs), 10)], reps=2)
        t_bulk, (rb, ro) = best(lambda: R.serialize_many(), reps=3)
        print(json.dumps({"what": "download", "dataset": name, "results": int(len(lhs)), "bytes": int(rb.size),
                          "per_bitmap_ms_extrapolated": t_one * 10 * 1e3, "bulk_ms": t_bulk * 1e3,
                          "bulk_GBps": rb.size / t_bulk / 1e9}), flush=True)
        t_fz, (fb, fo, fl) = best(lambda: R.frozen_serialize_many(), reps=3)
        t_fl, F = best(lambda: eng.pool_from_frozen(fb, fo, fl), reps=3)
        assert np.array_equal(F.cardinalities(), R.cardinalities())
        print(json.dumps({"what": "frozen", "dataset": name, "results": int(len(lhs)), "bytes": int(fb.size),
                          "serialize_ms": t_fz * 1e3, "serialize_GBps": fb.size / t_fz / 1e9, "load_ms": t_fl * 1e3,
                          "load_GBps": fb.size / t_fl / 1e9, "portable_bulk_ms": t_bulk * 1e3}), flush=True)
        F.free()
        t_opt, Q = best(lambda: eng.run_optimize(R), reps=3)
        print(json.dumps({"what": "run_optimize", "dataset": name, "containers": int(R.n_containers),
                          "payload_in": int(R.payload_bytes()), "payload_out": int(Q.payload_bytes()),
                          "ms": t_opt * 1e3}), flush=True)
        t_pred, eq = best(lambda: eng.pairwise_predicate("is_subset", H, lhs, H, rhs), reps=3)
        print(json.dumps({"what": "is_subset", "dataset": name, "pairs": int(len(lhs)), "true": int(eq.sum()),
                          "ms": t_pred * 1e3, "Mpairs_per_s": len(lhs) / t_pred / 1e6}), flush=True)
        ids = np.arange(0, len(bufs), 2, dtype=np.uint32)
        H2 = eng.pool_from_serialized(bufs)
        t_inp, _ = best(lambda: eng.pairwise_inplace("or", H2, ids, H2, ids + 1), reps=3)
        print(json.dumps({"what": "or_inplace", "dataset": name, "updates": int(ids.size), "ms": t_inp * 1e3}), flush=True)
    # C2-sized download / upload: 8 bitmaps x 4096 bitset containers = 256 MiB
    P = eng.pool_synth_bitset(8, 4096, 1)
    t_bulk, (rb, ro) = best(lambda: P.serialize_many(), reps=2)
    print(json.dumps({"what": "download", "dataset": "synthetic 8x4096 bitset", "bytes": int(rb.size),
                      "bulk_ms": t_bulk * 1e3, "bulk_GBps": rb.size / t_bulk / 1e9}), flush=True)
    t_dev, P2 = best(lambda: eng.pool_from_blob(rb, ro), reps=2)
    bufs = [rb[int(ro[i]):int(ro[i + 1])].tobytes() for i in range(8)]
    t_host, H = best(lambda: eng.pool_from_serialized(bufs), reps=1)
    assert np.array_equal(P2.cardinalities(), P.cardinalities())
    print(json.dumps({"what": "load", "dataset": "synthetic 8x4096 bitset", "bytes": int(rb.size),
                      "host_parser_ms": t_host * 1e3, "device_parser_ms": t_dev * 1e3,
                      "device_GBps": rb.size / t_dev / 1e9}), flush=True)


if __name__ == "__main__":
    main()
