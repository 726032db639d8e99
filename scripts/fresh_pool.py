"""What one C2 call costs when it has to allocate (and place) its result pool: RHIP_ARENA_HOLD / RHIP_ARENA_TRIES from the
environment; argv: ops.  Prints per op: ms of the fresh call, ms of the same call with `reuse`, probe rates."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import croaring_amd
eng = croaring_amd.Engine(0)
pool = eng.pool_synth_bitset(256, 4096, 0x9E3779B97F4A7C15)
k = np.arange(250, dtype=np.uint64)
lhs, rhs = (k % 256).astype(np.uint32), ((k * 97 + 1) % 256).astype(np.uint32)
for op in sys.argv[1:] or ["and", "or"]:
    t = time.perf_counter(); r = eng.pairwise(op, pool, lhs, pool, rhs); t_fresh = time.perf_counter() - t
    pl = eng.last_placement()
    ts = []
    for _ in range(5):
        t = time.perf_counter(); r = eng.pairwise(op, pool, lhs, pool, rhs, reuse=r); ts.append(time.perf_counter() - t)
    print(op, "fresh ms", round(t_fresh * 1e3, 1), "reuse ms", round(min(ts) * 1e3, 2), "hold", os.environ.get("RHIP_ARENA_HOLD"),
          "tries", os.environ.get("RHIP_ARENA_TRIES"), "probes", pl, flush=True)
    r.free()
