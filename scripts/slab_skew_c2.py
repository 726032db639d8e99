"""k_bb on C2 with the result arena at a chosen offset inside a 64 GiB allocation (RHIP_ARENA_TRIES=0: no search):
argv = offsets in GiB.  One fresh result pool per offset, and + or."""
import ctypes as C, json, os, sys
os.environ["RHIP_ARENA_TRIES"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import croaring_amd, bench
G = 1 << 30
eng = croaring_amd.Engine(0); eng.set_timing(True)
eng.lib.rhip_debug_set_arena_skew.argtypes = [C.c_void_p, C.c_ulonglong]; eng.lib.rhip_debug_set_arena_skew.restype = None
pool = eng.pool_synth_bitset(256, 4096, bench.SEED)
for off in [float(x) for x in sys.argv[1:]] or [0, 27, 28]:
    eng.lib.rhip_debug_set_arena_skew(eng.h, int(off * G))
    row = {"offset_GiB": off}
    for op in ("and", "or"):
        res, ks = None, []
        for i in range(4):
            lhs, rhs = bench.schedule(i * 250, 250, 256)
            res = eng.pairwise(op, pool, lhs, pool, rhs, reuse=res)
            ks.append(eng.last_stats()["ms_bitset_kernel"])
        row[op] = round(min(ks[1:]), 3)
        del res
    print(json.dumps(row), flush=True)
