"""rocprofv3 target / timing: the 200-way union of the C5 roaring64 bitmaps (rhip_or_many), 30 calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import croaring_amd
from util import c5_inputs
eng = croaring_amd.Engine(0)
bufs = c5_inputs()
pool = eng.pool_from_serialized64(bufs)
ts = []
for it in range(30):
    t = time.perf_counter()
    r = eng.or_many(pool)
    ts.append(time.perf_counter() - t)
    if it == 0:
        card = int(r.cardinalities()[0])
print("c5 union of", len(bufs), "MANY_DICT", os.environ.get("RHIP_MANY_DICT", "1"), "min ms", round(min(ts[3:]) * 1e3, 4), "median", round(float(np.median(ts[3:])) * 1e3, 4), "card", card)
