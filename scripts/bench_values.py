"""Value lists <-> pools on one MI355X (JSON lines): build the weather / census1881 pools from their decoded value
lists, decode all-pairs OR results back to values."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import croaring_amd
from util import all_pairs, load_bundle


def best(fn, reps=4):
    ts, out = [], None
    for _ in range(reps):
        t = time.perf_counter(); out = fn(); ts.append(time.perf_counter() - t)
    return min(ts), out


eng = croaring_amd.Engine()
for name in ("weather_sept_85", "census1881"):
    bufs = load_bundle(name)
    H = eng.pool_from_serialized(bufs)
    vals, offs = H.to_values()
    t_build, P = best(lambda: eng.pool_from_packed_values(vals, offs))
    t_opt, Q = best(lambda: eng.run_optimize(P))
    assert Q.serialize_all() == bufs
    print(json.dumps({"what": "of_ptr + run_optimize", "dataset": name, "values": int(vals.size), "build_ms": t_build * 1e3,
                      "run_optimize_ms": t_opt * 1e3, "Gvalues_per_s": vals.size / (t_build + t_opt) / 1e9}), flush=True)
    lhs, rhs = all_pairs(len(bufs))
    R = eng.pairwise("or", H, lhs, H, rhs)
    t_dec, (rv, ro) = best(lambda: R.to_values(), reps=3)
    print(json.dumps({"what": "to_uint32_array of all-pairs or", "dataset": name, "results": int(len(lhs)),
                      "values": int(rv.size), "ms": t_dec * 1e3, "Gvalues_per_s": rv.size / t_dec / 1e9,
                      "GBps_out": rv.size * 4 / t_dec / 1e9}), flush=True)
