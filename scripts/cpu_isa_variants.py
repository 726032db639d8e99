"""CPU baseline across the reference's ISA variants: CRoaring built as shipped (runtime dispatch:
AVX-512 where the host has it), with AVX-512 compiled out (AVX2), and with AVX compiled out (scalar), one core.
Needs `make -C oracle ref_isa` (build container only; the .so files travel to the GPU box).  JSON lines."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from gen_inputs import splitmix64
from oracle.pyoracle import Ref
from util import all_pairs, load_bundle

VARIANTS = [("default dispatch", "libcroaring_ref.so"), ("no AVX-512 (AVX2)", "libcroaring_ref_noavx512.so"),
            ("no AVX (scalar)", "libcroaring_ref_noavx.so")]


def rate(chk, hs, lhs, rhs, op, budget):
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget:
        for i, j in zip(lhs, rhs):
            r = chk.op(op, hs[i], hs[j]); chk.cardinality(r); chk.free(r); n += 1
            if n % 64 == 0 and time.perf_counter() - t0 > budget:
                break
    return n / (time.perf_counter() - t0)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
    flags = [l for l in open("/proc/cpuinfo") if l.startswith("flags")][:1]
    print(json.dumps({"host_has_avx512f": bool(flags and " avx512f" in flags[0]), "host_has_avx2": bool(flags and " avx2" in flags[0]),
                      "cpus": os.cpu_count()}), flush=True)
    NC = 4096
    words = [splitmix64((bench.SEED + b) & (2**64 - 1), NC * 1024) for b in range(4)]
    c2_bufs = [bench.portable_bitset_bitmap(w) for w in words]
    wbufs = load_bundle("weather_sept_85")
    wl, wr = all_pairs(len(wbufs))
    for label, fname in VARIANTS:
        path = os.path.join(ROOT, "oracle", "_ref", fname)
        if not os.path.exists(path):
            print(json.dumps({"variant": label, "error": "not built"}), flush=True)
            continue
        chk = type("RefVariant", (Ref,), {"PATH": path})()
        hs = [chk.deserialize(b) for b in c2_bufs]
        row = {"variant": label, "cores": 1}
        for op in ("and", "or"):
            r = rate(chk, hs, [0, 1, 2, 3], [1, 2, 3, 0], op, budget)
            row[f"c2_{op}_ops_per_s"] = round(r, 1)
            row[f"c2_{op}_GBps"] = round(r * NC * 24576 / 1e9, 2)
        for h in hs:
            chk.free(h)
        hs = [chk.deserialize(b) for b in wbufs]
        for op in ("and", "or"):
            row[f"weather_{op}_ops_per_s"] = round(rate(chk, hs, wl, wr, op, budget), 0)
        for h in hs:
            chk.free(h)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
