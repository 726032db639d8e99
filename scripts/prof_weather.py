"""rocprofv3 target: all-pairs batches of one realdata set (argv[2], default weather_sept_85; "c5" = the roaring64
configuration), one op per run (argv[1]), 12 timed batches; argv[3] == "pipe": two calls in flight (begin / end)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import croaring_amd
from util import load_bundle, all_pairs
op = sys.argv[1] if len(sys.argv) > 1 else "and"
name = sys.argv[2] if len(sys.argv) > 2 else "weather_sept_85"
eng = croaring_amd.Engine(0)
if name == "c5":  # BASELINE config C5: roaring64, wikileaks-noquotes x 10 high-32 buckets
    from util import c5_inputs
    bufs = c5_inputs()
    pool = eng.pool_from_serialized64(bufs)
else:
    bufs = load_bundle(name)
    pool = eng.pool_from_serialized(bufs)
lhs, rhs = all_pairs(len(bufs))
res = None
ts = []
if len(sys.argv) > 3 and sys.argv[3] == "pipe":
    slots, prev = [None, None], None
    for it in range(8 + 24):
        if it == 8:
            eng.host_clock(True)
            t0 = time.perf_counter()
        cur = eng.pairwise_begin(op, pool, lhs, pool, rhs, reuse=slots[it & 1])
        slots[it & 1] = None
        if prev is not None:
            slots[(it - 1) & 1] = prev.end()
        prev = cur
    slots[1] = prev.end()
    dt = (time.perf_counter() - t0) / 24
    print(op, name, "pipelined min ms", dt * 1e3, "host us/batch:", [round(x / 24, 1) for x in eng.host_clock()][:6])
    sys.exit(0)
plist = eng.pairlist_all_pairs(pool) if os.environ.get("LIST", "0") == "1" else None  # LIST=1: prepared pair list
for it in range(12):
    if it == 2:
        eng.host_clock(True)
    t = time.perf_counter()
    res = eng.pairwise_list(op, plist, reuse=res) if plist is not None else eng.pairwise(op, pool, lhs, pool, rhs, reuse=res)
    ts.append(time.perf_counter() - t)
hc = [round(x / 10, 1) for x in eng.host_clock()]
print(op, name, "min ms", min(ts) * 1e3, "mean(last 10) ms", sum(ts[2:]) / 10 * 1e3,
      "host us/batch [pairs, scratch+h2d, plan launches, class launches, wait, bookkeeping]:", hc[:6])
