"""rocprofv3 target: the device loader on C4's images (100 000 sparse bitmaps, 1.64 GB): rhip_pool_from_blob four times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import croaring_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
eng = croaring_amd.Engine(0)
blob, offs = croaring_amd.synth_sparse_portable(0, 1, n)
ts = []
for _ in range(4):
    t = time.perf_counter()
    pool = eng.pool_from_blob(blob, offs)
    ts.append(time.perf_counter() - t)
    pool.free()
print("loader", n, "bytes", int(blob.size), "ms per load", [round(x * 1e3, 2) for x in ts])
