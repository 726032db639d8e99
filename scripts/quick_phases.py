"""Phase breakdown of the wave-per-pair kernels (needs a diagnostic build: RHIP_EXTRA_FLAGS=-DRHIP_PHASES
python -m croaring_amd.build).  Prints, per workload and op, the share of wave time spent in each phase of
k_filter / k_wave (100 MHz ticks summed over all waves)."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import croaring_amd
from oracle.pyoracle import Oracle
from util import load_bundle, all_pairs

o = Oracle()
eng = croaring_amd.Engine(0)
eng.lib.rhip_debug_phases.restype = C.c_int
eng.lib.rhip_debug_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
rng = np.random.default_rng(3)
NB, NC = 48, 64
FILT = ["item", "zero+scatterX", "test+emit", "meta"]
WAVE = ["item", "buildX", "applyY", "reduce+type", "table", "emit", "meta"]


def phases(reset=True):
    buf = (C.c_ulonglong * 32)()
    assert eng.lib.rhip_debug_phases(eng.h, buf, 1 if reset else 0) == 0
    return np.array(list(buf), dtype=np.float64)


def pool(kind, card):
    bufs = []
    for _ in range(NB):
        parts = []
        for k in range(NC):
            if kind == "array":
                v = np.sort(rng.choice(65536, int(card * rng.uniform(0.5, 1.5)), replace=False))
            else:
                v = np.flatnonzero(rng.random(65536) < 0.5)
            parts.append((np.uint32(k) << np.uint32(16)) | v.astype(np.uint32))
        h = o.from_sorted(np.concatenate(parts), run_optimize=False)
        bufs.append(o.serialize(h)); o.free(h)
    return eng.pool_from_serialized(bufs)


def report(tag, pa, lhs, pb, rhs, items):
    for op in ("and", "or"):
        eng.pairwise(op, pa, lhs, pb, rhs)
        phases()
        t = time.perf_counter()
        eng.pairwise(op, pa, lhs, pb, rhs)
        dt = time.perf_counter() - t
        p = phases()
        row = {"workload": tag, "op": op, "call_ms": round(dt * 1e3, 3)}
        for name, base, labels in (("k_filter", 0, FILT), ("k_wave", 8, WAVE)):
            tot = p[base:base + 8].sum()
            if tot > 0:
                row[name] = {"wave_us_per_item": round(tot / 100.0 / max(items, 1), 3),
                             "share": {l: round(float(p[base + i] / tot), 3) for i, l in enumerate(labels)}}
        print(json.dumps(row), flush=True)


pools = {"A200": pool("array", 200), "A874": pool("array", 874), "B": pool("bitset", 0)}
lhs, rhs = np.meshgrid(np.arange(NB, dtype=np.uint32), np.arange(NB, dtype=np.uint32))
lhs, rhs = lhs.ravel().copy(), rhs.ravel().copy()
for a, b in (("A200", "A200"), ("A874", "A874"), ("A874", "B")):
    report(f"{a} x {b}", pools[a], lhs, pools[b], rhs, lhs.size * NC)
bufs = load_bundle("weather_sept_85")
P = eng.pool_from_serialized(bufs)
l, r = all_pairs(len(bufs))
report("weather_sept_85", P, l, P, r, 223255)
