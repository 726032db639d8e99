"""Quick realdata timing (no CPU baseline): all-pairs, 4 ops, min of 7 batches, one JSON line per (dataset, op)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import croaring_amd
from util import load_bundle, all_pairs, OPS
eng = croaring_amd.Engine(0)
for name in sys.argv[1:] or ["weather_sept_85", "census1881", "census-income", "wikileaks-noquotes"]:
    bufs = load_bundle(name)
    pool = eng.pool_from_serialized(bufs)
    lhs, rhs = all_pairs(len(bufs))
    row = {"dataset": name}
    for op in OPS:
        res, ts = None, []
        for _ in range(8):
            t = time.perf_counter()
            res = eng.pairwise(op, pool, lhs, pool, rhs, reuse=res)
            ts.append(time.perf_counter() - t)
        st = eng.last_stats()
        row[op] = {"ms": round(min(ts[1:]) * 1e3, 4), "Mops": round(len(lhs) / min(ts[1:]) / 1e6, 2),
                   "TBps": round((st["bytes_in"] + st["bytes_out"]) / min(ts[1:]) / 1e12, 3)}
        # two calls in flight (begin / end): per-call period over 30 calls
        slots, prev = [res, None], None
        for it in range(8 + 30):  # 8 warm-up calls: every slot's pinned staging exists before the clock starts
            if it == 8:
                t0 = time.perf_counter()
            cur = eng.pairwise_begin(op, pool, lhs, pool, rhs, reuse=slots[it & 1])
            slots[it & 1] = None
            if prev is not None:
                slots[(it - 1) & 1] = prev.end()
            prev = cur
        slots[1] = prev.end()
        row[op]["ms_pipelined2"] = round((time.perf_counter() - t0) / 30 * 1e3, 4)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); eng.pairwise_cardinality("and", pool, lhs, pool, rhs); t.append(time.perf_counter() - t0)
    row["and_card_ms"] = round(min(t[1:]) * 1e3, 4)
    print(json.dumps(row), flush=True)
