"""Where inside ONE large allocation does a C2 result arena stream fast?  Pool = the C2 operand pool (8 GiB); one slab of
SLAB GiB (argv[1], default 64); the placement probe of an 8 GiB arena at every STEP GiB (argv[2], default 1) inside it.
One line per process: run it several times (fresh processes) to see whether the fast offsets repeat."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import croaring_amd, bench
G = 1 << 30
slab = float(sys.argv[1]) if len(sys.argv) > 1 else 64
step = float(sys.argv[2]) if len(sys.argv) > 2 else 1
need = float(sys.argv[3]) if len(sys.argv) > 3 else 8
eng = croaring_amd.Engine(0)
f = eng.lib.rhip_debug_probe_offsets
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_ulonglong, C.c_ulonglong, C.POINTER(C.c_float), C.c_int]
eng.lib.rhip_debug_pool_arena.argtypes = [C.c_void_p]; eng.lib.rhip_debug_pool_arena.restype = C.c_ulonglong
pool = eng.pool_synth_bitset(256, 4096, bench.SEED)
out = (C.c_float * 256)()
n = f(eng.h, pool.h, int(slab * G), int(need * G), int(step * G), out, 256)
r = [round(float(out[k])) for k in range(max(n, 0))]
print(json.dumps({"slab_GiB": slab, "step_GiB": step, "need_GiB": need, "pool_va_GiB": round(eng.lib.rhip_debug_pool_arena(pool.h) / G, 2), "n": n,
                  "GBps": r, "fast_offsets_GiB": [round(k * step, 2) for k, v in enumerate(r) if v >= 6200]}), flush=True)
