"""A/B helper: median and min of N synchronous all-pairs batches over a prepared pair list (argv: dataset op [op ...]); run it once
per library variant (RHIP_LIB_VARIANT), alternating, and compare the medians."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import croaring_amd
from util import load_bundle
eng = croaring_amd.Engine(0)
name = sys.argv[1]
pool = eng.pool_from_serialized(load_bundle(name))
plist = eng.pairlist_all_pairs(pool)
row = {}
for op in sys.argv[2:]:
    res, ts = None, []
    for _ in range(60):
        t = time.perf_counter()
        res = eng.pairwise_list(op, plist, reuse=res)
        ts.append(time.perf_counter() - t)
    ts = np.array(ts[10:]) * 1e3
    row[op] = [round(float(np.median(ts)), 4), round(float(ts.min()), 4)]
print(os.environ.get("RHIP_LIB_VARIANT", "") or "product", name, json.dumps(row), flush=True)
