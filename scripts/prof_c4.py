"""rocprofv3 target: C4 or_many over the seeded sparse bitmaps (argv[1] = number of bitmaps; above 100 000 the pool is
built in pieces, as bench.py's c4x10 row does), 6 calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import croaring_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
eng = croaring_amd.Engine(0)
if n > 100000:
    from bench import sparse_pool
    pool = sparse_pool(eng, 0, 1, n)
else:
    blob, offs = croaring_amd.synth_sparse_portable(0, 1, n)
    pool = eng.pool_from_blob(blob, offs)
ts = []
for _ in range(6):
    t = time.perf_counter()
    r = eng.or_many(pool)
    ts.append(time.perf_counter() - t)
print("c4 or_many", n, "min ms", min(ts) * 1e3, "all", [round(x * 1e3, 3) for x in ts])
