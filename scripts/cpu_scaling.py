"""How does the CPU reference scale over host cores on this box? (fork-based workers, shared inputs)"""
import os, sys, time, multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from gen_inputs import splitmix64
from oracle.pyoracle import Ref, Oracle
chk = Ref() if Ref.available() else Oracle()
NC = 4096
hs = [chk.deserialize(bench.portable_bitset_bitmap(splitmix64((bench.SEED + b) & (2**64 - 1), NC * 1024))) for b in range(8)]
lhs, rhs = bench.schedule(0, 1 << 20, 8)

def work(args):
    tid, T, secs = args
    done, k = 0, tid
    end = time.perf_counter() + secs
    while time.perf_counter() < end:
        for op in ("and", "or"):
            r = chk.op(op, hs[lhs[k]], hs[rhs[k]]); chk.cardinality(r); chk.free(r); done += 1
        k += T
    return done

print("cpu_count", os.cpu_count(), "checker", chk.name, flush=True)
for T in (1, 8, 16, 32, 64, 128, 256):
    if T > (os.cpu_count() or 1): break
    with mp.get_context("fork").Pool(T) as pool:
        t0 = time.perf_counter()
        res = pool.map(work, [(t, T, 4.0) for t in range(T)])
        dt = time.perf_counter() - t0
    print(f"T={T:4d}: {sum(res)/dt:9.1f} ops/s  ({sum(res)*NC*24576/dt/1e9:7.1f} GB/s algorithmic)", flush=True)
