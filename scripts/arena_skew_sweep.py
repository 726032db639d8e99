"""k_bb duration on C2 (250 pairs x 4096 bitset containers) against HOW the result arena is allocated: one pool, one
process, the result arena re-allocated for every row (size rounded up to a multiple of round_MiB, 0 = as is).
python scripts/arena_skew_sweep.py"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import croaring_amd, bench
M = 1 << 20
eng = croaring_amd.Engine(0); eng.set_timing(True)
eng.lib.rhip_debug_set_arena_round.argtypes = [C.c_void_p, C.c_ulonglong]; eng.lib.rhip_debug_set_arena_round.restype = None
eng.lib.rhip_debug_pool_arena.argtypes = [C.c_void_p]; eng.lib.rhip_debug_pool_arena.restype = C.c_ulonglong
pool = eng.pool_synth_bitset(256, 4096, bench.SEED)
pa = eng.lib.rhip_debug_pool_arena(pool.h)
for rnd in [0, 0, 0, 1024, 1024, 1024, 2048, 2048, 2048, 4096, 4096, 16384, 16384, 0, 1024, 16384]:
    eng.lib.rhip_debug_set_arena_round(eng.h, rnd * M)
    row = {"round_MiB": rnd}
    for op in ("and", "or"):
        res, ks = None, []
        for i in range(4):
            lhs, rhs = bench.schedule(i * 250, 250, 256)
            res = eng.pairwise(op, pool, lhs, pool, rhs, reuse=res)
            ks.append(eng.last_stats()["ms_bitset_kernel"])
        ra = eng.lib.rhip_debug_pool_arena(res.h)
        row[op] = round(min(ks[1:]), 3)
        row[op + "_addr_GiB"] = round(ra / (1 << 30), 4)
        del res
    print(json.dumps(row), flush=True)
