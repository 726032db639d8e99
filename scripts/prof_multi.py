"""rocprofv3 target: all-pairs rhip_pairwise_multi (and, or, xor, andnot) batches of one realdata set (argv[1]), 12 timed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import croaring_amd
from util import load_bundle, all_pairs, OPS
name = sys.argv[1] if len(sys.argv) > 1 else "census1881"
eng = croaring_amd.Engine(0)
if name == "c5":
    from util import c5_inputs
    bufs = c5_inputs(); pool = eng.pool_from_serialized64(bufs)
else:
    bufs = load_bundle(name); pool = eng.pool_from_serialized(bufs)
lhs, rhs = all_pairs(len(bufs))
res, ts = None, []
for it in range(12):
    if it == 2:
        eng.host_clock(True)
    t = time.perf_counter(); res = eng.pairwise_multi(list(OPS), pool, lhs, pool, rhs, reuse=res); ts.append(time.perf_counter() - t)
print("multi4", name, "min ms", min(ts) * 1e3, "host us/batch:", [round(x / 10, 1) for x in eng.host_clock()][:6])
