"""Compact realdata timing: all-pairs, 4 ops, min of 7 synchronous batches (ms) per dataset, one line per dataset.
argv: dataset names ("c5" = roaring64 wikileaks x 10)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import croaring_amd
from util import load_bundle, all_pairs, OPS
eng = croaring_amd.Engine(0)
tag = os.environ.get("TAG", "")
for name in sys.argv[1:] or ["weather_sept_85", "census1881", "census-income", "wikileaks-noquotes", "c5"]:
    if name == "c5":
        from util import c5_inputs
        bufs = c5_inputs(); pool = eng.pool_from_serialized64(bufs)
    else:
        bufs = load_bundle(name); pool = eng.pool_from_serialized(bufs)
    lhs, rhs = all_pairs(len(bufs))
    plist = eng.pairlist_all_pairs(pool) if os.environ.get("LIST", "0") == "1" else None  # LIST=1: prepared pair list
    row = {}
    for op in OPS:
        res, ts = None, []
        for _ in range(9):
            t = time.perf_counter()
            res = eng.pairwise_list(op, plist, reuse=res) if plist is not None else eng.pairwise(op, pool, lhs, pool, rhs, reuse=res)
            ts.append(time.perf_counter() - t)
        row[op] = round(min(ts[2:]) * 1e3, 4)
    print(tag, name, json.dumps(row), flush=True)
    if os.environ.get("MULTI", "1") == "1":
        res, ts = None, []
        for _ in range(9):
            t = time.perf_counter()
            res = eng.pairwise_list(list(OPS), plist, reuse=res) if plist is not None else eng.pairwise_multi(list(OPS), pool, lhs, pool, rhs, reuse=res)
            ts.append(time.perf_counter() - t)
        print(tag, name, "multi4 ms", round(min(ts[2:]) * 1e3, 4), "sum of singles", round(sum(row.values()), 4), flush=True)
