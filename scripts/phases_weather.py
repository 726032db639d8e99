"""Per-phase share of wave time in k_filter / k_wave / k_usmall on weather_sept_85 all-pairs (diagnostic build:
RHIP_EXTRA_FLAGS=-DRHIP_PHASES python -m croaring_amd.build; 100 MHz ticks summed over all waves)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import croaring_amd
from util import load_bundle, all_pairs
eng = croaring_amd.Engine(0)
eng.lib.rhip_debug_phases.restype = C.c_int
eng.lib.rhip_debug_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
NAMES = {"k_filter": (0, ["item", "stage X", "test+emit", "meta"]),
         "k_wave": (8, ["item", "build X", "apply Y", "reduce+type", "table", "emit", "meta"]),
         "k_usmall": (16, ["item+zero", "probe ranks", "stream X", "new values+meta"]),
         "k_genw": (24, ["item", "interval path", "image A", "image B", "combine+count+type", "emit"])}
def phases():
    buf = (C.c_ulonglong * 32)()
    assert eng.lib.rhip_debug_phases(eng.h, buf, 1) == 0
    return np.array(list(buf), dtype=np.float64)
pool = eng.pool_from_serialized(load_bundle(sys.argv[1] if len(sys.argv) > 1 else "weather_sept_85"))
lhs, rhs = all_pairs(len(pool))
for op in ("and", "or", "andnot") if len(sys.argv) < 3 else sys.argv[2:]:
    eng.pairwise(op, pool, lhs, pool, rhs)
    phases()
    for _ in range(4):
        eng.pairwise(op, pool, lhs, pool, rhs)
    ph = phases()
    for k, (base, names) in NAMES.items():
        v = ph[base:base + len(names)]
        if v.sum() > 0:
            print(op, k, "wave-ms %.1f" % (v.sum() / 1e5 / 4), {n: "%.0f%%" % (100 * x / v.sum()) for n, x in zip(names, v)})
