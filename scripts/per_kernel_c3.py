"""Per-kernel algorithmic GB/s of the class kernels on a realdata all-pairs batch: rhip_last_class_stats (items, payload
bytes in / out per class) next to the STAND-ALONE duration of each kernel (RHIP_NO_OVERLAP=1 in the environment:
one stream, no co-running) taken with HIP events around... no -- taken from rocprofv3's kernel trace of this very
process (argv[2] = the trace directory is read afterwards by the caller); this script prints the class statistics.
Usage: RHIP_NO_OVERLAP=1 rocprofv3 --kernel-trace ... -- python scripts/per_kernel_c3.py <dataset>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import croaring_amd
from util import load_bundle, all_pairs, OPS
name = sys.argv[1] if len(sys.argv) > 1 else "weather_sept_85"
eng = croaring_amd.Engine(0)
if name == "c5":
    from util import c5_inputs
    pool = eng.pool_from_serialized64(c5_inputs())
else:
    pool = eng.pool_from_serialized(load_bundle(name))
lhs, rhs = all_pairs(len(pool))
for op in OPS:
    res = None
    for _ in range(6):  # (kernel durations: averaged over these by the caller)
        res = eng.pairwise(op, pool, lhs, pool, rhs, reuse=res)
    eng.set_class_stats(True)
    eng.pairwise(op, pool, lhs, pool, rhs)
    eng.set_class_stats(False)
    print(json.dumps({"dataset": name, "op": op, "classes": {k: v for k, v in eng.last_class_stats().items() if v["items"]}}), flush=True)
    eng.synchronize()
    print("MARK", op, flush=True)
