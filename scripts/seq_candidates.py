"""Probe rate of N consecutive exact-size allocations (all kept alive) against the C2 operand pool: which positions of a
sequential allocation run land in a fast window?  argv: N (default 14), size GiB (default 8)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
import croaring_amd, bench
G = 1 << 30
N = int(sys.argv[1]) if len(sys.argv) > 1 else 14
need = float(sys.argv[2]) if len(sys.argv) > 2 else 8
eng = croaring_amd.Engine(0)
f = eng.lib.rhip_debug_probe_offsets
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_ulonglong, C.c_ulonglong, C.POINTER(C.c_float), C.c_int]
pool = eng.pool_synth_bitset(256, 4096, bench.SEED)
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
keep, rates = [], []
# (the probe entry point allocates its own slab and frees it: here every candidate must stay alive, so allocate through
# torch and probe in place with a slab the size of the candidate -- offsets = [0])
import torch
out = (C.c_float * 4)()
g = eng.lib.rhip_debug_probe_at
g.restype = C.c_float
g.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulonglong]
for k in range(N):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), int(need * G)) == 0
    keep.append(p)
    rates.append(round(g(eng.h, pool.h, p, int(need * G))))
print(json.dumps({"need_GiB": need, "GBps": rates, "va_GiB": [round((k.value - keep[0].value) / G, 1) for k in keep]}), flush=True)
